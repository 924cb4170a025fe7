"""Latent video transformer meta-architecture (reference: vidgen/modeling/meta_arch/vt.py:22-328).

supervised: one random subscale slice per sample -> mean_k CE(logits_k, slice_k | ignore first N_PRIME frames).
inference : logits for an entire video (BitsEvaluator) and / or autoregressive sampling (VTSampler).
"""
import os

import torch
from torch import nn

from ...engine.grad_reducer import BucketedGradReducer
from ...hip import binding as L
from ...hip import tx
from ...solver import build_lr_scheduler, build_optimizer
from ...utils.checkpoint import Checkpointer
from ...utils.events import get_event_storage
from ..autoregressive import build_autoregressive
from ..autoregressive.vt_attention import prefetch_p2_images
from ..autoregressive.vt_utils import slice_and_context, subscale_order
from .build import META_ARCH_REGISTRY
from .common import init_weights, stack_to_device


class _XentFn(torch.autograd.Function):
    """scale * mean over non-ignored positions of CE(logits_tok (rows, nv), target[:, k])."""

    @staticmethod
    def forward(ctx, logits, target, k, ignore, scale):
        b, nc, P = target.shape
        loss, lse, count = tx.xent_fwd(logits, target[0, k], nc * P, 1, P, ignore, scale)
        ctx.save_for_backward(logits, target, lse, count)
        ctx.args = (k, ignore, scale)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target, lse, count = ctx.saved_tensors
        k, ignore, scale = ctx.args
        b, nc, P = target.shape
        return tx.xent_bwd(logits, target[0, k], nc * P, 1, P, ignore, lse, count, g.contiguous().view(1), scale), \
            None, None, None, None


# groups of a large sampling batch run on their own streams (False: one after the other on the caller's stream;
# the results are identical -- tests/test_gpu_sampling.py)
DECODE_GROUP_STREAMS = os.environ.get("LVT_DECODE_GROUP_STREAMS", "1") != "0"      # "0": groups one after the other
MAX_CONCURRENT_GROUPS = 3
DECODE_GROUP_ROWS = 256          # videos per decode group (rows of every decode-step launch)
# The host issues a decode step (one graph launch per group) ~50x faster than the GPU executes it and nothing in the
# sampling loop needs a result on the host, so unchecked it would queue every step of every remaining slice (thousands of
# graph launches, ~10^6 AQL packets across streams that wait on each other) ahead of the device.  The loop therefore never
# runs more than this many positions ahead of the slowest group (event per window, host waits on the window before last).
DECODE_MAX_STEPS_AHEAD = int(os.environ.get("LVT_DECODE_MAX_STEPS_AHEAD", "64"))


class _RunAhead:
    """Bounds how far the host may run ahead of the streams it feeds: tick() after every position; every `window`
    positions an event is recorded on each stream and the host waits for the events of the window before last."""

    def __init__(self, streams, max_ahead):
        self.streams, self.window = streams, max(1, max_ahead // 2)
        self.count, self.pending = 0, []

    def tick(self):
        self.count += 1
        if self.count % self.window:
            return
        evs = []
        for s in self.streams:
            e = torch.cuda.Event()
            e.record(s)
            evs.append(e)
        self.pending.append(evs)
        if len(self.pending) > 1:
            for e in self.pending.pop(0):
                e.synchronize()

    def drain(self):
        for evs in self.pending:
            for e in evs:
                e.synchronize()
        self.pending = []

@META_ARCH_REGISTRY.register()
class VideoTransformerModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.model = build_autoregressive(cfg)
        self.init_weights(self.model, cfg.MODEL.INIT_TYPE)
        self.vis_period = cfg.VIS_PERIOD
        self._reducers = []
        self.to(self.device)

    init_weights = staticmethod(init_weights)

    def train(self, mode=True):
        self.training = mode
        self.model.train(mode)
        return self

    def wrap_parallel(self, device_ids, broadcast_buffers):
        """Bucketed RCCL gradient averaging overlapped with backward (reference: torch DDP, vt.py:61-63).
        Self-joining like DDP: the reducer hooks `Optimizer.step`, so the reference loop
        (engine/trainer.py:79-87: forward, backward, step) needs no extra call."""
        for r in self._reducers:
            r.remove()
        self._reducers = [BucketedGradReducer(self.model.parameters())]

    def finish_gradient_sync(self):
        """Optional early join for callers that read `.grad` before `optimizer.step()`."""
        for r in self._reducers:
            r.wait()

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise L.LvtError("lvt_amd models compute only on a MI355X (MODEL.DEVICE=%s); there is no CPU path"
                             % self.device)

    @property
    def _vt(self):
        return self.cfg.MODEL.AUTOREGRESSIVE.VT

    # ---- supervised ------------------------------------------------------------------------------------
    def preprocess_data(self, data):
        """list[dict] -> (context, slice, slice_idx, ignore_mask, class_idx) on the device (vt.py:284-299)."""
        self._require_gpu()
        ctx = stack_to_device([x["context"] for x in data], self.device)
        sl = stack_to_device([x["slice"] for x in data], self.device)
        sidx = stack_to_device([x["slice_idx"] for x in data], self.device)
        ign = stack_to_device([x["ignore_mask"] for x in data], self.device)
        return ctx, sl, sidx, ign, self._class_idx(data)

    def _class_idx(self, data):
        """(b,) int64 class labels when the samples carry them (vt.py:216-219, 235-238, 294-297)."""
        if "class" not in data[0]:
            return None
        return stack_to_device([torch.as_tensor(x["class"]).long() for x in data], self.device)

    def _begin_pass(self):
        """f16x2 arithmetic: the max |.| records of the parameters are per pass (hip/binding.py) -- forget the old ones and
        scan every weight matrix in one launch.  A no-op in the other math modes."""
        L.bump_epoch()
        L.prefetch_module_weights(self)
        prefetch_p2_images(self)

    def compute_supervised_loss(self, context, slice, slice_idx, ignore_mask, iter=0, class_idx=None):
        self._begin_pass()
        ignore = self.cfg.MODEL.IGNORE_INDEX
        b, nc = slice.shape[:2]
        target = torch.masked_fill(slice, ignore_mask, ignore).reshape(b, nc, -1).contiguous()
        logits = self.model.logits_tokens(context.contiguous(), slice.contiguous(), slice_idx.contiguous(), class_idx)
        loss = 0
        for k in range(nc):
            loss = loss + _XentFn.apply(logits[k], target, k, ignore, 1.0 / nc)
        return {"loss_cross_entropy": loss}

    def forward(self, data, mode="inference"):
        if mode != "supervised":
            self._begin_pass()                   # (compute_supervised_loss begins its own)
        if mode == "supervised":
            self.finish_gradient_sync()      # accumulation: the previous micro-step's all-reduce owns the buckets
            context, slice, slice_idx, ignore_mask, class_idx = self.preprocess_data(data)
            it = get_event_storage().iter
            return self.compute_supervised_loss(context, slice, slice_idx, ignore_mask, it, class_idx)
        if mode == "inference":
            self._require_gpu()
            output = [{} for _ in range(len(data))]
            if "BitsEvaluator" in self.cfg.TEST.EVALUATORS:
                output = self.calculate_logits_for_entire_video(data, output)
            if "VTSampler" in self.cfg.TEST.EVALUATORS:
                output = self.sample_videos(data, output, n_prime=self.cfg.TEST.VT_SAMPLER.N_PRIME,
                                            num_samples=self.cfg.TEST.VT_SAMPLER.NUM_SAMPLES)
            assert len(output[0]) > 0
            return output
        raise ValueError("|mode| is invalid")

    # ---- likelihood of a whole video (vt.py:230-282) ------------------------------------------------------
    @torch.no_grad()
    def calculate_logits_for_entire_video(self, data, output):
        v = self._vt
        video = stack_to_device([torch.as_tensor(x["image_sequence"]) for x in data], self.device)
        B, T, nc, H, W = video.shape
        video = video.transpose(1, 2).contiguous()                   # B, nc, T, H, W
        class_idx = self._class_idx(data)
        st, sh, sw = v.STRIDE
        idx2abc, _ = subscale_order(st, sh, sw)
        t, h, w = T // st, H // sh, W // sw
        logits = torch.zeros(B, nc, v.NV, T, H, W, device=video.device)
        for si, (a, b_, c) in enumerate(idx2abc):
            sl, ctx = slice_and_context(video, a, b_, c, v.STRIDE, v.KERNEL, v.PAD_VALUE)
            sidx = torch.full((B,), si, dtype=torch.long, device=video.device)
            pred = self.model.logits_tokens(ctx.contiguous(), sl, sidx, class_idx)   # nc x (B*t*h*w, nv)
            for k in range(nc):
                logits[:, k, :, a::st, b_::sh, c::sw] = pred[k].view(B, t, h, w, v.NV).permute(0, 4, 1, 2, 3)
        ignore_mask = torch.zeros(1, T, H, W, dtype=torch.bool, device=video.device)
        if v.N_PRIME > 0:
            ignore_mask[:, :v.N_PRIME] = True
        for i in range(B):
            output[i]["ignore_mask"] = ignore_mask
            output[i]["logits"] = logits[i]
        return output

    # ---- sampling (vt.py:82-136, 210-228) -----------------------------------------------------------------
    @torch.no_grad()
    def sample_video(self, video, temp=1.0, n_prime=1, class_idx=None, incremental=True):
        """video (B, nc, T, H, W) int64 with the first n_prime frames given; returns the completed grid.

        incremental=True (default): encoder once per slice, then ONE single-token decoder step per position
        against K/V caches (modeling/autoregressive/incremental.py).  incremental=False reproduces the
        reference's schedule (full decoder pass per generated pixel, vt.py:121-131); both draw from the same
        per-pixel distributions."""
        from ..autoregressive.incremental import GraphedSliceSampler
        self._require_gpu()
        v = self._vt
        video = video.to(self.device).clone()
        st, sh, sw = v.STRIDE
        idx2abc, _ = subscale_order(st, sh, sw)
        B, nc, T, H, W = video.shape
        t, h, w = T // st, H // sh, W // sw
        prime = torch.zeros(T, H, W, dtype=torch.bool)
        if n_prime > 0:
            prime[:n_prime] = True
        pred = self.model.ch_predictor
        sampler = None
        # the K/V-cache decoder attends over the whole slice; block-split layers (slice larger than the attention
        # block, e.g. DSSVT at 16 frames) fall back to the reference schedule
        if any(tuple(l.block_size) != (t, h, w) for l in self.model.decoder.block_local_attention):
            incremental = False
        groups = None
        if incremental:
            # A decode step is ~90 dependent launches of a few dozen workgroups each: latency bound, most CUs idle.
            # Two levers fill the chip: more videos per launch (the decode kernels take one workgroup per 64 rows, so a
            # group of 256 videos is 4x the workgroups at almost the same step time), and independent groups on separate
            # streams (own K/V caches and graphs) whose steps interleave on the GPU.  Measured on 1 MI355X, frames/s:
            # 527 (64 videos), 932 (128), 1341 (256), 1696 (512) in one group; 1956 for 3 groups of 256.
            ng = (B + DECODE_GROUP_ROWS - 1) // DECODE_GROUP_ROWS
            bounds = [B * g // ng for g in range(ng + 1)]
            key = (B, t, h, w, float(temp))
            groups = self._samplers.get(key) if hasattr(self, "_samplers") else None
            if groups is None:
                groups = [(bounds[g], bounds[g + 1], GraphedSliceSampler(self.model, bounds[g + 1] - bounds[g], (t, h, w), temp),
                           (torch.cuda.Stream(device=video.device) if DECODE_GROUP_STREAMS else
                            torch.cuda.current_stream(video.device)) if ng > 1 else None) for g in range(ng)]
                self._samplers = {key: groups}            # keep the most recent geometry's graphs
            sampler = groups[0][2]
        for si, (a, b_, c) in enumerate(idx2abc):
            sl, ctx = slice_and_context(video, a, b_, c, v.STRIDE, v.KERNEL, v.PAD_VALUE)
            prime_sl = prime[a::st, b_::sh, c::sw]
            if bool(prime_sl.all()):
                continue
            sidx = torch.full((B,), si, dtype=torch.long, device=video.device)
            zl = self.model.encoder.forward_tokens(ctx.contiguous(), sidx, class_idx)   # context is fixed per slice
            if sampler is not None:
                flat = prime_sl.reshape(-1).tolist()
                S = t * h * w
                if len(groups) == 1:
                    sampler.begin_slice(zl, sl)
                    ahead = _RunAhead([torch.cuda.current_stream(video.device)], DECODE_MAX_STEPS_AHEAD)
                    for pos in range(S):
                        sampler.step(pos, sample=not flat[pos])     # primed pixels only fill the K/V caches
                        ahead.tick()
                    sl = sampler.sl.clone()
                else:
                    main = torch.cuda.current_stream(video.device)
                    zl3 = zl.view(B, S, -1)
                    # at most MAX_CONCURRENT_GROUPS groups at a time: measured 527 / 873 / 1165 frames/s for 1 / 2 / 3
                    # concurrent groups of 64 and a collapse to ~520-860 with four or more (any GPU_MAX_HW_QUEUES)
                    nwaves = (len(groups) + MAX_CONCURRENT_GROUPS - 1) // MAX_CONCURRENT_GROUPS
                    for wv in range(nwaves):                    # balanced: 4 groups run as 2 + 2, not 3 + 1
                        wave = groups[len(groups) * wv // nwaves:len(groups) * (wv + 1) // nwaves]
                        for g0, g1, smp, stream in wave:
                            stream.wait_stream(main)
                            with torch.cuda.stream(stream):
                                smp.begin_slice(zl3[g0:g1].reshape((g1 - g0) * S, -1), sl[g0:g1])
                        for _, _, _, stream in wave:            # every group starts after ALL slice set-ups (shared tables)
                            for _, _, _, other in wave:
                                if other is not stream:
                                    stream.wait_stream(other)
                        ahead = _RunAhead([g[3] for g in wave], DECODE_MAX_STEPS_AHEAD)
                        for pos in range(S):
                            for g0, g1, smp, stream in wave:
                                with torch.cuda.stream(stream):
                                    smp.step(pos, sample=not flat[pos])
                            ahead.tick()
                        for g0, g1, smp, stream in wave:
                            with torch.cuda.stream(stream):
                                sl[g0:g1] = smp.sl
                            main.wait_stream(stream)
            else:
                for ti in range(t):
                    for hi in range(h):
                        for wi in range(w):
                            if prime_sl[ti, hi, wi]:
                                continue
                            yl = self.model.decoder.forward_tokens(sl, zl)
                            sl[:, :, ti, hi, wi] = pred.sample_pixel_tokens(yl, B, t * h * w, (ti * h + hi) * w + wi, temp)
            video[:, :, a::st, b_::sh, c::sw] = sl
        return video

    @torch.no_grad()
    def sample_slice(self, context, slice_idx, slice_size, temp=0.9, class_idx=None):
        sl = torch.zeros(size=slice_size, device=context.device, dtype=torch.long)
        B, _, t, h, w = sl.shape
        zl = self.model.encoder.forward_tokens(context.contiguous(), slice_idx.contiguous(), class_idx)
        for ti in range(t):
            for hi in range(h):
                for wi in range(w):
                    yl = self.model.decoder.forward_tokens(sl, zl)
                    sl[:, :, ti, hi, wi] = self.model.ch_predictor.sample_pixel_tokens(
                        yl, B, t * h * w, (ti * h + hi) * w + wi, temp)
        return sl

    def sample_videos(self, data, output, n_prime=5, num_samples=1):
        video = stack_to_device([torch.as_tensor(x["image_sequence"]) for x in data], self.device)
        video = video.transpose(1, 2).contiguous()                   # B, nc, T, H, W
        video[:, :, n_prime:] = 0
        class_idx = self._class_idx(data)
        samples = [self.sample_video(video.clone(), n_prime=n_prime, class_idx=class_idx) for _ in range(num_samples)]
        assert video.size(0) == len(output)
        for i in range(video.size(0)):
            output[i]["samples"] = [s[i] for s in samples]
        return output

    # ---- optimizers / checkpointers (vt.py:316-328) -----------------------------------------------------
    def configure_optimizers_and_checkpointers(self):
        opt = build_optimizer(self.model, self.cfg, suffix="_G")
        os.makedirs(os.path.join(self.cfg.OUTPUT_DIR, "netG"), exist_ok=True)
        c = [{"checkpointer": Checkpointer(self.model, os.path.join(self.cfg.OUTPUT_DIR, "netG")),
              "pretrained": self.cfg.MODEL.GENERATOR.WEIGHTS}]
        o = [{"optimizer": opt, "scheduler": build_lr_scheduler(self.cfg, opt), "type": "generator"}]
        return o, c
