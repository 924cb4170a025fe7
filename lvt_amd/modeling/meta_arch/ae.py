"""Auto-encoder meta-architecture (reference: vidgen/modeling/meta_arch/ae.py:21-244).

Same constructor, modes, method names and checkpoint layout; compute is the HIP conv stack on
channels-last activations.  There is no CPU path: a forward on a non-GPU device raises."""
import os

import numpy as np
import torch
from torch import nn

from ...engine.grad_reducer import BucketedGradReducer
from ...hip import binding as L
from ...hip import ew
from ...solver import build_lr_scheduler, build_optimizer
from ...utils.checkpoint import Checkpointer
from ...utils.events import get_event_storage
from .. import convstack
from ..encoder import build_encoder
from ..generator import build_generator
from ..loss.loss import mse
from .build import META_ARCH_REGISTRY
from .common import init_weights, stack_to_device


@META_ARCH_REGISTRY.register()
class AutoEncoderModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.encoder = build_encoder(cfg)
        self.init_weights(self.encoder, cfg.MODEL.INIT_TYPE)
        self.generator = build_generator(cfg)
        self.init_weights(self.generator, cfg.MODEL.INIT_TYPE)
        assert len(cfg.MODEL.PIXEL_MEAN) == len(cfg.MODEL.PIXEL_STD)
        c = len(cfg.MODEL.PIXEL_MEAN)
        # kept as plain tensors like the reference (lambdas over captured tensors, ae.py:32-37)
        self._pixel_mean = torch.tensor(cfg.MODEL.PIXEL_MEAN, dtype=torch.float32, device=self.device)
        self._pixel_std = torch.tensor(cfg.MODEL.PIXEL_STD, dtype=torch.float32, device=self.device)
        self.normalizer = lambda x: (x - self._pixel_mean.view(1, c, 1, 1)) / self._pixel_std.view(1, c, 1, 1)
        self.back_normalizer = lambda y: y * self._pixel_std.view(1, c, 1, 1) + self._pixel_mean.view(1, c, 1, 1)
        self.vis_period = cfg.VIS_PERIOD
        self._reducers = []
        self.to(self.device)

    init_weights = staticmethod(init_weights)

    # ---- engine-facing contract (ae.py:63-84) ------------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        self.encoder.train(mode)
        self.generator.train(mode)
        return self

    def wrap_parallel(self, device_ids, broadcast_buffers):
        """Data-parallel gradient averaging for encoder and generator.  The reference wraps each
        sub-network in torch DDP; here a bucketed RCCL all-reduce is hooked onto the parameters and
        runs on a side stream while the rest of the backward is still computing.  Self-joining like DDP:
        the reducers hook `Optimizer.step`, so the reference loop (engine/trainer.py:79-87) needs no extra call."""
        for r in self._reducers:
            r.remove()
        self._reducers = [BucketedGradReducer(self.encoder.parameters()),
                          BucketedGradReducer(self.generator.parameters())]

    def finish_gradient_sync(self):
        """Optional early join for callers that read `.grad` before `optimizer.step()`."""
        for r in self._reducers:
            r.wait()

    def _generator_parameters(self):
        return list(self.encoder.parameters()) + list(self.generator.parameters())

    def _set_requires_grad(self, params, requires_grad):
        for p in params:
            p.requires_grad = requires_grad

    def set_generator_requires_grad(self, requires_grad):
        self._set_requires_grad(self._generator_parameters(), requires_grad)

    # ---- data staging (ae.py:151-168) -------------------------------------------------------------
    def _require_gpu(self):
        if self.device.type != "cuda":
            raise L.LvtError("lvt_amd models compute only on a MI355X (MODEL.DEVICE=%s); there is no CPU path"
                             % self.device)

    def _preprocess_cl(self, data):
        """list[dict] -> (normalised channels-last frames (N,1,H,W,C4), clip shape (b,t) or None)."""
        self._require_gpu()
        if "image" in data[0]:
            raw, bt = stack_to_device([x["image"] for x in data], self.device), None
        elif "image_sequence" in data[0]:
            raw = stack_to_device([x["image_sequence"] for x in data], self.device)
            bt = tuple(raw.shape[:2])
            raw = raw.view(-1, *raw.shape[2:])
        else:
            raise ValueError
        raw = raw.float().contiguous()
        n, c, h, w = raw.shape
        cp = (c + 3) // 4 * 4
        x = ew.to_channels_last(raw.view(n, c, h * w), cp, 1, self._pixel_mean, self._pixel_std)
        return x.view(n, 1, h, w, cp), bt

    def preprocess_data(self, data):
        """Reference contract: normalised (B,C,H,W) or (B,T,C,H,W) tensor."""
        x, bt = self._preprocess_cl(data)
        y = convstack.cl_to_nchw(x, len(self.cfg.MODEL.PIXEL_MEAN))
        return y if bt is None else y.view(*bt, *y.shape[1:])

    # ---- forward modes (ae.py:101-149) ------------------------------------------------------------
    def forward(self, data, mode="inference"):
        L.bump_epoch()                       # f16x2 arithmetic: max |.| records of parameters are per pass (hip/binding.py)
        L.prefetch_module_weights(self)
        if mode in ("generator", "supervised"):
            self.finish_gradient_sync()      # accumulation: the previous micro-step's all-reduce owns the buckets
            x, _ = self._preprocess_cl(data)
            if mode == "generator":
                it = get_event_storage().iter
                if self.vis_period > 0 and it > 0 and it % self.vis_period == 0:
                    with torch.no_grad():
                        self.visualize_training(x)
                return self._generator_loss_cl(x)
            return self._supervised_loss_cl(x)
        if mode == "encoder":
            x, bt = self._preprocess_cl(data)
            z = self._encode_cl(x)
            return z if bt is None else z.view(*bt, *z.shape[1:])
        if mode == "encoder_decoder":
            x, bt = self._preprocess_cl(data)
            out = convstack.cl_to_nchw(self._decode_cl(self._latent_cl(x)), self.generator.out_channels)
            return out if bt is None else out.view(*bt, *out.shape[1:])
        if mode == "interpolate_first_last":
            return self.interpolate_first_last(self.preprocess_data(data))
        if mode == "inference":
            x, bt = self._preprocess_cl(data)
            latent = self._latent_cl(x)
            y = self._decode_cl(latent)
            n, _, h, w, cp = y.shape
            hi = 1.0 if self.cfg.INPUT.SCALE_TO_ZEROONE else 255.0
            out = ew.to_channels_first(y.view(n, h * w, cp), self.generator.out_channels, 2, self._pixel_mean,
                                       self._pixel_std, 0.0, hi).view(n, -1, h, w)
            latent = self._latent_public(latent)
            if bt is not None:
                out = out.view(*bt, *out.shape[1:])
                latent = latent.view(*bt, *latent.shape[1:])
            return [{"reconstruction": out[i], "latent": latent[i]} for i in range(out.size(0))]
        raise ValueError("|mode| is invalid")

    # ---- channels-last internals (overridden by VQVAEModel) -----------------------------------------
    def _latent_cl(self, x):
        return self.encoder.forward_cl(x)

    def _latent_public(self, latent):
        return convstack.cl_to_nchw(latent, self.encoder.out_channels)

    def _encode_cl(self, x):
        return self._latent_public(self._latent_cl(x))

    def _decode_cl(self, latent):
        return self.generator.forward_cl(latent)

    def _generator_loss_cl(self, x):
        out = self.generator.forward_cl(self.encoder.forward_cl(x))
        c = len(self.cfg.MODEL.PIXEL_MEAN)
        return {"loss_ae_mse": mse(out, x, denom=x.numel() // x.shape[-1] * c)}

    def _supervised_loss_cl(self, x):
        return self._generator_loss_cl(x)

    # ---- the reference's tensor-level methods, (N,C,H,W) contract (ae.py:170-222) -----------------
    def _as_cl(self, x):
        self._require_gpu()
        if x.dim() == 5:
            x = x.reshape(-1, *x.shape[2:])
        return convstack.nchw_to_cl(x.contiguous().float())

    def compute_generator_loss(self, x):
        return self._generator_loss_cl(self._as_cl(x))

    def compute_supervised_loss(self, x):
        return self._supervised_loss_cl(self._as_cl(x))

    def encode(self, x):
        z = self._encode_cl(self._as_cl(x))
        return z.view(*x.shape[:2], *z.shape[1:]) if x.dim() == 5 else z

    def decode(self, latent):
        return convstack.cl_to_nchw(self.generator.forward_cl(convstack.nchw_to_cl(latent.contiguous())),
                                    self.generator.out_channels)

    def encode_decode(self, x, return_latent=False):
        latent = self._latent_cl(self._as_cl(x))
        out = convstack.cl_to_nchw(self._decode_cl(latent), self.generator.out_channels)
        latent = self._latent_public(latent)
        if x.dim() == 5:
            out = out.view(*x.shape[:2], *out.shape[1:])
            latent = latent.view(*x.shape[:2], *latent.shape[1:])
        return (out, latent) if return_latent else out

    def interpolate_first_last(self, x):
        b = x.size(0)
        if x.dim() == 5:
            return torch.stack([self.interpolate_first_last(x[i]) for i in range(b)], dim=0)
        alphas = torch.tensor(np.linspace(0, 1, b)).to(self.device).view(b, 1, 1, 1).float()
        start, end = self.encoder(x[0].unsqueeze(0)), self.encoder(x[-1].unsqueeze(0))
        return self.generator(self.lerp(start, end, alphas))

    @staticmethod
    def lerp(start, end, weights):
        return start + weights * (end - start)

    def visualize_training(self, x):
        storage = get_event_storage()
        recon = convstack.cl_to_nchw(self._decode_cl(self._latent_cl(x[:3])), self.generator.out_channels)
        for img in recon:
            storage.put_image("reconstruction", img.detach().cpu().numpy())

    # ---- optimizers / checkpointers (ae.py:224-244) -------------------------------------------------
    def configure_optimizers_and_checkpointers(self):
        opt_e = build_optimizer(self.encoder, self.cfg, suffix="_G")
        opt_g = build_optimizer(self.generator, self.cfg, suffix="_G")
        for sub in ("netE", "netG"):
            os.makedirs(os.path.join(self.cfg.OUTPUT_DIR, sub), exist_ok=True)
        c = [{"checkpointer": Checkpointer(self.encoder, os.path.join(self.cfg.OUTPUT_DIR, "netE")),
              "pretrained": self.cfg.MODEL.ENCODER.WEIGHTS},
             {"checkpointer": Checkpointer(self.generator, os.path.join(self.cfg.OUTPUT_DIR, "netG")),
              "pretrained": self.cfg.MODEL.GENERATOR.WEIGHTS}]
        o = [{"optimizer": opt_e, "scheduler": build_lr_scheduler(self.cfg, opt_e), "type": "generator"},
             {"optimizer": opt_g, "scheduler": build_lr_scheduler(self.cfg, opt_g), "type": "generator"}]
        return o, c
