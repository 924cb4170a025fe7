#!/usr/bin/env python
"""Headline benchmark (BASELINE.json `metric`): video-clips/s/node of the VQVAE+DSFVT train step on synthetic
BAIR-shaped clips (64x64x16, fp32) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: spawns one rank per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A STEP takes 64 clips per GPU through both models of the metric, with inputs already resident in HBM:
two PR-DVQVAE2 train steps (fwd + bwd + Adam) of 32 clips x 16 frames each (BASELINE configs[1]: IMS_PER_BATCH 32) and
one DSFVT train step (fwd + bwd + RMSprop) on 64 slices, one random subscale slice per clip (configs[2]/[3]:
IMS_PER_BATCH 64).  `value` = 64 * N * K / (time of exactly K such steps between two barriers, max over ranks).
One process per GPU, gradients averaged by a bucketed RCCL all-reduce overlapped with backward, EMA statistics by
one fused all-reduce; weak scaling.  Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      the matrix-core engine launches of the step (conv / GEMM / fused attention kernels), event-timed per
                launch on the launch stream in a SECOND pass of the same K steps (a timing event is a barrier packet;
                inside the timed region it costs ~15 % of a step)
  cpu_baseline  the CPU oracle (port of the reference's PyTorch-CPU path) on the host cores, bounded samples
  parity        the gates BASELINE.md section 3 asks for next to a throughput number: codebook rows of the timed data
                below the fp64 margin rule, and index flips / loss error against the CPU oracle on a sub-batch
  comm          world size seen by RCCL, all-reduce bytes per step, step time with the reducers switched off
  extra         each leg alone (own timed region, own roofline), the LVT_MATH=f32 variants, VQ-encode fractions and
                the end-to-end generation figure (configs[4]; run in a child process, see bench_generate)
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E
CLIP_FRAMES = 16
SEED = 29871897                    # configs/*/Base: SEED


def engine_peak(mode, kind=""):
    """Ceiling of an engine launch in ALGORITHMIC fp32 FLOP/s: the fp32 MFMA peak in 'f32' mode; in 'bf16x3' mode every
    algorithmic product is six bf16 MFMA products (a1b1 a1b2 a2b1 a1b3 a3b1 a2b2), so the ceiling is bf16 peak / 6; in
    'f16x2' mode three fp16 MFMA products (hi hi, hi lo, lo hi; fp16 rate = bf16 rate), so peak / 3 -- the attention launches
    too since round 5 (csrc/attention_flash.hip; with LVT_NO_FLASH_ATTENTION they fall back to the bf16x3 plane kernels)."""
    if mode == "f32":
        return FP32_MFMA_PEAK_TFLOPS
    if mode == "f16x2" and not (kind.startswith("attn_") and os.environ.get("LVT_NO_FLASH_ATTENTION")):
        return BF16_MFMA_PEAK_TFLOPS / 3.0
    return BF16_MFMA_PEAK_TFLOPS / 6.0


MATH_NOTE = {
    "f32": "fp32 in, fp32 out, v_mfma_f32_32x32x2_f32 products, fp32 accumulation",
    "bf16x3": "fp32 in, fp32 out, fp32 accumulation; every fp32 product formed exactly from a 3-way bf16 split of both "
              "operands (six v_mfma_f32_32x32x16_bf16 per block, dropped terms < 2^-24 |a||b|); measured error against "
              "fp64 is not larger than the plain fp32 MFMA path's (profiles/r01_math_mode_accuracy.txt, "
              "tests/test_gpu_engine.py::test_math_modes_accuracy); LVT_MATH=f32 selects the plain fp32 instruction",
    "f16x2": "fp32 in, fp32 out, fp32 accumulation; every fp32 operand is scaled by an exact power of two taken from its "
             "max |.| and split into two fp16 terms (hi + 2^-11 lo: 22 bits + sign), a block is three "
             "v_mfma_f32_32x32x16_f16 (hi hi | hi lo + lo hi in a second accumulator; dropped lo lo <= 2^-22 |a||b|, "
             "2^-24.6 rms); measured error against fp64 is not larger than the plain fp32 MFMA path's on seven operand "
             "classes (tests/test_gpu_engine.py::test_math_modes_accuracy) and every parity test runs in this mode at "
             "unchanged tolerances; the flash attention kernels split per ROW (token x head) instead of per tensor; the VQ search "
             "runs argmax(x.e - |e|^2/2) on the same two fp16 planes with one scale per row and one per codebook group "
             "(lvt_vq_nearest_f16x2_kernel, 0 of 524,288 indices differ from an fp64 search); the q/k/v and first-FFN forward "
             "products take both operands as ready fp16 planes staged by LDS-DMA (csrc/gemm_p2.hip, bit-identical)",
}
PEAK_NOTE = {
    "f32": "dense fp32 MFMA peak (MI355X_MICROARCH.md)",
    "bf16x3": "dense bf16 MFMA peak 2516.6 TFLOP/s / 6 MFMA products per algorithmic fp32 product; `achieved` counts "
              "ALGORITHMIC fp32 FLOPs only (not the 6x executed bf16 FLOPs); an MFMA-only loop on random bf16 operands "
              "sustains 71 % of that peak on this part (power throttling by operand toggling, "
              "profiles/r01_ubench_engine_bounds.txt), i.e. ~300 TFLOP/s in these units; the shader clock measured inside "
              "lvt_gemm_kernel on random operands is 1.58-1.69 GHz of the nominal 2.4 (2.25 GHz on zeros), "
              "profiles/r03_gemm_shape_and_power_probes.txt",
    "f16x2": "dense bf16/fp16 MFMA peak 2516.6 TFLOP/s / 3 MFMA products per algorithmic fp32 product; `achieved` counts "
             "ALGORITHMIC fp32 FLOPs only (the attention backward is counted with the reference's four products, 8 B H S^2 d_a, "
             "although the flash kernels recompute the scores and dP in both of their launches)",
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dp-single-rank", action="store_true",
                    help="with --gpus 1: join a ONE-rank RCCL group and keep the gradient reducers active (all-reduces are "
                         "identities there), so that `comm` reports the exposed communication of the data-parallel path")
    ap.add_argument("--rccl-channels", type=int, default=0,
                    help="cap RCCL's channels (NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS = n, set before the process group is made): "
                         "every channel of an all-reduce is a workgroup that needs a CU while the backward pass runs one engine "
                         "workgroup per CU; 0 = RCCL's default.  Reported under comm.rccl_channels")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batches", type=int, default=4, help="distinct synthetic batches rotated through the steps")
    ap.add_argument("--batch-clips", type=int, default=32, help="clips per GPU per VQ-VAE train step (BASELINE: 32)")
    ap.add_argument("--dsfvt-batch", type=int, default=64, help="slices (= clips) per GPU per DSFVT train step (BASELINE: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the per-leg timed regions under `extra`")
    ap.add_argument("--no-strict-f32", action="store_true", help="skip the secondary LVT_MATH=f32 figures")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gates")
    ap.add_argument("--no-generate", action="store_true", help="skip the end-to-end generation figure (BASELINE configs[4])")
    ap.add_argument("--generate-batch", type=int, default=768,
                    help="videos generated at once (decoded as groups of <= 256 on separate streams)")
    ap.add_argument("--generate-only", action="store_true", help="(child process of the generation leg) print its JSON and exit")
    ap.add_argument("--generate-timeout", type=float, default=240.0)
    ap.add_argument("--generate-cpu-baseline", action="store_true",
                    help="also time the reference's sampling schedule on the host cores (bounded sample, extrapolated)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="budget of the CPU baseline samples (whole run)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launch: `python bench.py --gpus N` starts its own ranks (reference: vidgen/engine/launch.py:25-67, tools/train_net.py:94-104)
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(local_rank, world, port, argv):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    run(parse(argv))


def self_launch(args, argv):
    """One process per GPU on this node, rendezvous on 127.0.0.1 at a free port; rank 0 prints the JSON line."""
    import torch.multiprocessing as mp
    mp.spawn(_rank_main, args=(args.gpus, free_port(), list(argv)), nprocs=args.gpus, join=True)


# ---------------------------------------------------------------------------------------------------------------------
# timing helpers
# ---------------------------------------------------------------------------------------------------------------------
def _pct(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    pos = q * (len(xs) - 1)
    lo = int(pos)
    hi = min(lo + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (pos - lo)


def step_stats(events):
    """Per-step durations from HIP events recorded on the launch stream at the step boundaries (one event per step:
    tens of ms apart, so the barrier packet an event inserts is not measurable)."""
    ms = [events[i].elapsed_time(events[i + 1]) for i in range(len(events) - 1)]
    return {"median_ms": round(_pct(ms, 0.5), 3), "p10_ms": round(_pct(ms, 0.1), 3), "p90_ms": round(_pct(ms, 0.9), 3),
            "n": len(ms)}


def timed_steps(step, steps, first_iter, world, device):
    """The timed region of the contract: barrier + synchronize, EXACTLY `steps` steps, synchronize + barrier; MAX over
    ranks.  Returns (seconds, per-step stats, last step's result)."""
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    events = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    events[0].record()
    out = None
    for i in range(steps):
        out = step(first_iter + i)
        events[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item()), step_stats(events), out


def traffic_from_profile(name, launches_per_step=None):
    """HBM bytes per engine launch from this round's committed PMC passes (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
    separate runs of tools/profile/bench_leg.py, summarised by tools/profile/pmc_summary.py / tools/profile/round_profiles.sh): counters cannot be
    read inside the run.  -> (bytes per launch or None, file name, stale?): the figure is STALE when the profiled code issued
    another number of engine launches per step than the run that quotes it (the kernels changed since the pass)."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
    except Exception:
        return None, None, None
    # `engine_calls_per_step`: engine wrapper calls of one step of the profiled code, counted by tools/profile/bench_leg.py exactly as
    # the KernelTimer of this run counts them (a wrapper call may be several kernel dispatches, which is what the counters see)
    prof = d.get("engine_calls_per_step")
    stale = None if (prof is None or launches_per_step is None) else bool(abs(prof - launches_per_step) > 0.5)
    return round(d["hbm_bytes_per_launch"]), name, stale


def engine_summary(timer, steps, mode):
    summ = timer.summary()
    eng = {k: v for k, v in summ.items() if k.startswith("conv_") or k.startswith("gemm_") or k.startswith("attn_")}
    tot_ms = sum(v["ms"] for v in eng.values())
    tot_fl = sum(v["flops"] for v in eng.values())
    launches = sum(v["launches"] for v in eng.values())
    achieved = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    per_kind = {k: {"launches_per_step": v["launches"] // steps, "avg_us": round(v["ms"] / v["launches"] * 1e3, 1),
                    "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                    "ms_per_step": round(v["ms"] / steps, 3)} for k, v in sorted(eng.items())}
    other = {k: {"launches_per_step": v["launches"] // steps, "avg_us": round(v["ms"] / v["launches"] * 1e3, 1),
                 "ms_per_step": round(v["ms"] / steps, 3)} for k, v in sorted(summ.items()) if k not in eng}
    # time the launches would take at their own ceilings / measured time (= achieved / peak when all share one ceiling)
    ideal_ms = sum(v["flops"] / (engine_peak(mode, k) * 1e12) * 1e3 for k, v in eng.items())
    peak = tot_fl / (ideal_ms * 1e-3) / 1e12 if ideal_ms > 0 else engine_peak(mode)
    return {"achieved": achieved, "ms_per_step": tot_ms / steps, "launches_per_step": launches // steps,
            "flops_per_launch": tot_fl / max(launches, 1), "per_kind": per_kind, "frac": achieved / peak, "peak": peak,
            "other": other}


def instrumented(step, steps, first_iter):
    """`steps` more steps with a HIP event pair around every engine launch -> (engine summary input, ms per step)."""
    from lvt_amd.hip import binding as L
    L.TIMER = L.KernelTimer()
    t1 = time.perf_counter()
    for i in range(steps):
        step(first_iter + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) / steps * 1e3
    timer, L.TIMER = L.TIMER, None
    return timer, ms


def roofline_block(es, mode, traffic_file, kernel_note):
    traffic, src, stale = traffic_from_profile(traffic_file, es["launches_per_step"])
    return {"bound": "mfma", "achieved": round(es["achieved"], 2), "peak": round(es["peak"], 1), "unit": "TFLOP/s",
            "frac": round(es["frac"], 4), "traffic": traffic, "traffic_stale": stale,
            "traffic_unit": "HBM bytes per engine launch (rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE in "
                            "separate passes of this command, profiles/%s); algorithmic flops per launch = %.3e"
                            % (src, es["flops_per_launch"]),
            "kernel": kernel_note, "peak_note": PEAK_NOTE[mode], "per_kind": es["per_kind"]}


# ---------------------------------------------------------------------------------------------------------------------
# the two legs of the metric
# ---------------------------------------------------------------------------------------------------------------------
class VqvaeLeg:
    """PR-DVQVAE2 train step: fwd + bwd + Adam on `batch` clips x 16 frames (encoder, 4x512 EMA codebooks, decoder)."""

    def __init__(self, device, world, rank, local_rank, batch, nbatches, dp=None):
        from lvt_amd.config import get_cfg
        from lvt_amd.modeling import build_model
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(ROOT, "configs/vqvae/PR-DVQVAE2.yaml"))
        cfg.MODEL.DEVICE = device
        cfg.OUTPUT_DIR = "/tmp/lvt_bench_out"
        torch.manual_seed(SEED + rank)
        self.cfg, self.model = cfg, build_model(cfg)
        self.model.train()
        self.optimizers, _ = self.model.configure_optimizers_and_checkpointers()
        if world > 1 if dp is None else dp:
            self.model.wrap_parallel(device_ids=[local_rank], broadcast_buffers=False)
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        self.clips, self.batches = [], []
        for _ in range(nbatches):                   # synthetic clips, resident in HBM before any timed region
            clips = torch.rand(batch, CLIP_FRAMES, 3, 64, 64, generator=g).to(device)
            self.clips.append(clips)
            self.batches.append([{"image_sequence": clips[i]} for i in range(batch)])
        self.batch, self.nbatches = batch, nbatches

    def step(self, i):
        from lvt_amd.utils.events import EventStorage
        with EventStorage(i):
            losses = self.model(self.batches[i % self.nbatches], mode="supervised")
        sum(losses.values()).backward()
        for o in self.optimizers:     # (under data parallelism the gradient all-reduce joins itself before step)
            o["optimizer"].step()
        for o in self.optimizers:
            o["optimizer"].zero_grad()
        return losses


class DsfvtLeg:
    """DSFVT train step: fwd + bwd + RMSprop on `batch` samples = one random subscale slice (256 tokens x 4 code channels)
    of one 16-frame code clip each."""

    def __init__(self, device, world, rank, local_rank, batch, nbatches, dp=None):
        from lvt_amd.config import get_cfg
        from lvt_amd.data.dataset_mapper import prepare_slices_batch
        from lvt_amd.modeling import build_model
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(ROOT, "configs/vt/DSFVT.yaml"))
        cfg.MODEL.DEVICE = device
        cfg.OUTPUT_DIR = "/tmp/lvt_bench_out"
        torch.manual_seed(SEED + rank)
        self.cfg, self.model = cfg, build_model(cfg)
        self.model.train()
        self.optimizers, _ = self.model.configure_optimizers_and_checkpointers()
        if world > 1 if dp is None else dp:
            self.model.wrap_parallel(device_ids=[local_rank], broadcast_buffers=False)
        v = cfg.MODEL.AUTOREGRESSIVE.VT
        g = torch.Generator(device="cpu").manual_seed(4321 + rank)
        self.batches = []
        for _ in range(nbatches):                  # distinct clips and slice offsets per batch, built on the device
            codes = torch.randint(0, v.NV, (batch, 16, v.NC, 16, 16), generator=g).to(device)
            abcs = [(int(a), 0, 0) for a in torch.randint(v.N_PRIME, 16, (batch,), generator=g)]
            self.batches.append(prepare_slices_batch(codes, abcs, v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE))
        self.batch, self.nbatches = batch, nbatches

    def step(self, i):
        from lvt_amd.utils.events import EventStorage
        ctx, sl, sidx, ign = self.batches[i % self.nbatches]
        with EventStorage(i):
            loss = self.model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
        loss.backward()                            # (the gradient all-reduce joins itself before optimizer.step)
        for o in self.optimizers:
            o["optimizer"].step()
        for o in self.optimizers:
            o["optimizer"].zero_grad()
        return loss


def host_fed(vq, ds, steps, world, device):
    """The same bench step fed from HOST memory the way a training loop feeds the models: per-sample numpy arrays in list[dict]
    (the reference's DataLoader output; ae.py:151-168, vt.py:284-299), staged by lvt_amd.data.prefetch.DevicePrefetcher -- one
    pinned buffer and one asynchronous H2D copy per key and batch, issued while the previous step computes -- and handed to
    `model(data, mode="supervised")`.  2 x 25.2 MB (PR-DVQVAE2, 32 clips) + 4.3 MB (DSFVT, 64 slices) cross PCIe per step.
    Reported beside `value` (whose inputs are resident in HBM), never as `value`."""
    import itertools
    import numpy as np
    from lvt_amd.data.prefetch import DevicePrefetcher
    from lvt_amd.utils.events import EventStorage
    vq_host = [[{"image_sequence": c.cpu().numpy()} for c in clips] for clips in vq.clips]
    ds_host = []
    for ctx, sl, sidx, ign in ds.batches:
        c, s_, i_, g_ = ctx.cpu().numpy(), sl.cpu().numpy(), sidx.cpu().numpy(), ign.cpu().numpy()
        ds_host.append([{"context": c[j], "slice": s_[j], "slice_idx": i_[j], "ignore_mask": g_[j]} for j in range(c.shape[0])])
    n_vq = max(1, ds.batch // vq.batch)              # VQ-VAE train steps per DSFVT train step (2)

    def run(n):
        vq_it = iter(DevicePrefetcher(itertools.islice(itertools.cycle(vq_host), n * n_vq), device))
        ds_it = iter(DevicePrefetcher(itertools.islice(itertools.cycle(ds_host), n), device))
        for i in range(n):
            for j in range(n_vq):
                with EventStorage(i):
                    losses = vq.model(next(vq_it), mode="supervised")
                sum(losses.values()).backward()
                for o in vq.optimizers:
                    o["optimizer"].step()
                for o in vq.optimizers:
                    o["optimizer"].zero_grad()
            with EventStorage(i):
                loss = ds.model(next(ds_it), mode="supervised")["loss_cross_entropy"]
            loss.backward()
            for o in ds.optimizers:
                o["optimizer"].step()
            for o in ds.optimizers:
                o["optimizer"].zero_grad()
    run(2)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h2d = (n_vq * vq.batch * CLIP_FRAMES * 3 * 64 * 64 * 4 + sum(v.nbytes for v in ds_host[0][0].values() if hasattr(v, "nbytes")) * ds.batch)
    return {"clips_per_s": round(ds.batch * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "h2d_bytes_per_step": int(h2d),
            "note": "per-GPU rate of the bench step with every batch coming from host memory as list[dict] of per-sample numpy "
                    "arrays through DevicePrefetcher (pinned staging, asynchronous copy on a side stream, overlapped with the "
                    "previous step) and model(data, mode='supervised'); PCIe-inclusive, reported beside `value`, never as it"}


def leg_alone(name, leg, steps, warmup, world, device, units, traffic_file, kernel_note, strict_f32):
    """One leg in its own timed region + instrumented pass -> the `extra.<leg>` block."""
    from lvt_amd.hip import binding as L
    for i in range(warmup):
        leg.step(i)
    elapsed, stats, _ = timed_steps(leg.step, steps, warmup, world, device)
    timer, inst_ms = instrumented(leg.step, steps, warmup + steps)
    mode = L.get_math_mode()
    es = engine_summary(timer, steps, mode)
    out = {"%s_per_s" % units: round(leg.batch * world * steps / elapsed, 2), "ms_per_step": round(elapsed / steps * 1e3, 3),
           "step_ms": stats, "steps": steps, "warmup": warmup, "batch_per_gpu": leg.batch,
           "engine_ms_per_step": round(es["ms_per_step"], 3), "engine_launches_per_step": es["launches_per_step"],
           "roofline": roofline_block(es, mode, traffic_file, kernel_note), "other_kernels": es["other"]}
    if strict_f32 and mode != "f32":
        L.set_math_mode("f32")
        for i in range(2):
            leg.step(i)
        n2 = max(3, steps // 3)
        e2, _, _ = timed_steps(leg.step, n2, 2, world, device)
        timer2, _ = instrumented(leg.step, n2, 2 + n2)
        L.set_math_mode(mode)
        s2 = engine_summary(timer2, n2, "f32")
        out["strict_f32_mfma"] = {"%s_per_s" % units: round(leg.batch * world * n2 / e2, 2), "ms_per_step": round(e2 / n2 * 1e3, 3),
                                  "engine_tflops": round(s2["achieved"], 2), "mfma_utilisation": round(s2["frac"], 4),
                                  "note": "LVT_MATH=f32: the identical step on v_mfma_f32_32x32x2_f32; utilisation = fraction "
                                          "of the 157.3 TFLOP/s fp32 MFMA peak over the engine launches"}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# parity gates and the VQ-encode fractions
# ---------------------------------------------------------------------------------------------------------------------
def vq_gates(vq, device, oracle_clips):
    """On the first timed batch with the weights the timed steps left behind: (i) count of codebook searches whose fp64
    top-2 margin is below 1e-5 (|x|^2 + max|e|^2) (the rule of tests/util_models.py:margin_ok -- rows the reference's own
    fp32 summation order decides); (ii) the HIP search against an fp64 search on the device: differing rows, all of which
    must be sub-margin; (iii) the event-timed VQ encode.  Returns (gate dict, state for the CPU-oracle comparison)."""
    from lvt_amd.hip import binding as L
    model = vq.model
    model.eval()
    with torch.no_grad():
        x_cl, _ = model._preprocess_cl(vq.batches[0])
        z = model.encoder.forward_cl(x_cl)                              # (N,1,16,16,256) channels-last
        n = z.shape[0]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        idx = model.codebook.indices_cl(z)
        e0.record()
        for _ in range(reps):
            idx = model.codebook.indices_cl(z)                           # (N,num,16,16) int64
        e1.record()
        torch.cuda.synchronize()
        vq_us = e0.elapsed_time(e1) / reps * 1e3
        sd = {k: v.detach() for k, v in model.codebook.state_dict().items()}
        num = idx.shape[1]
        d = z.shape[-1] // num
        rows = z.view(-1, num, d)
        sub, differ, differ_sub = 0, 0, 0
        for i in range(num):
            e = sd["ve.%d.embedding.weight" % i].double()
            xi = rows[:, i].double()
            best = torch.empty(xi.shape[0], dtype=torch.int64, device=z.device)
            margin = torch.empty(xi.shape[0], dtype=torch.float64, device=z.device)
            for s in range(0, xi.shape[0], 65536):                       # fp64 distances, 64k rows at a time
                xs = xi[s:s + 65536]
                dd = (e ** 2).sum(1)[None] + (xs ** 2).sum(1, keepdim=True) - 2.0 * xs @ e.t()
                t2 = torch.topk(dd, 2, dim=1, largest=False)
                best[s:s + 65536] = t2.indices[:, 0]
                margin[s:s + 65536] = t2.values[:, 1] - t2.values[:, 0]
            ok = margin > 1e-5 * ((xi ** 2).sum(1) + (e ** 2).sum(1).max())
            mine = idx[:, i].reshape(-1)
            bad = mine != best
            sub += int((~ok).sum())
            differ += int(bad.sum())
            differ_sub += int((bad & ~ok).sum())
    model.train()
    frames = n
    alg_bytes = 270336.0 * frames                    # 262,144 B z_e read + 8,192 B int64 indices per frame
    alg_flops = 2.0 * 33.5e6 * frames                # 33.5 M MAC per frame
    mode = L.get_math_mode()
    gate = {"vq_searches_checked": int(rows.shape[0] * num), "sub_margin_rows": sub,
            "margin_rule": "fp64 top-2 distance gap <= 1e-5 * (|x|^2 + max|e|^2): rows the reference's own fp32 addmm order decides",
            "hip_vs_fp64_search_differing_rows": differ, "of_which_sub_margin": differ_sub,
            "bit_exact_on_clear_margin_rows": differ == differ_sub}
    enc = {"us_per_launch": round(vq_us, 1), "frames": frames,
           "hbm_frac": round(alg_bytes / (vq_us * 1e-6) / (HBM_PEAK_GBS * 1e9), 4),
           "hbm_gbs": round(alg_bytes / (vq_us * 1e-6) / 1e9, 1),
           "flop_frac": round(alg_flops / (vq_us * 1e-6) / 1e12 / engine_peak(mode, "vq_search"), 4),
           "tflops": round(alg_flops / (vq_us * 1e-6) / 1e12, 1),
           "note": "lvt_vq_nearest on the z_e of one timed batch (all four codebooks): algorithmic bytes 270,336 B/frame "
                   "against the 8 TB/s HBM peak and 33.5 M MAC/frame against the engine ceiling (SURVEY 8d: 248 FLOP/B)"}
    k = min(oracle_clips, vq.batch) * CLIP_FRAMES
    state = {"enc": {a: b.detach().cpu().clone() for a, b in model.encoder.state_dict().items()},
             "dec": {a: b.detach().cpu().clone() for a, b in model.generator.state_dict().items()},
             "cb": {a: b.detach().cpu().clone() for a, b in model.codebook.state_dict().items()},
             "x01": vq.clips[0][:min(oracle_clips, vq.batch)].reshape(k, 3, 64, 64).cpu(), "idx": idx[:k].cpu()}
    return gate, enc, state


def oracle_flips(state):
    """CPU oracle (restatement of the reference) on the same weights and the same <= 8 clips: index flips vs the HIP path."""
    from oracle import lvt_oracle as O
    with torch.no_grad():
        x = O.normalize(state["x01"], (0.5,) * 3, (0.5,) * 3)
        z_e = O.res_encoder(state["enc"], x)
        theirs = O.dvq_indices(state["cb"], z_e)
        flips = theirs != state["idx"]
        sub = 0
        num = theirs.shape[1]
        for i, part in enumerate(z_e.split(z_e.size(1) // num, dim=1)):
            rows = part.permute(0, 2, 3, 1).reshape(-1, part.shape[1])
            cb = state["cb"]["ve.%d.embedding.weight" % i]
            d0, d1, _ = O.vq_margin_fp64(rows, cb)
            ok = (d1 - d0) > 1e-5 * ((rows.double() ** 2).sum(1) + (cb.double() ** 2).sum(1).max())
            sub += int((flips[:, i].reshape(-1) & ~ok).sum())
    return {"oracle_clips": state["x01"].shape[0] // CLIP_FRAMES, "searches": int(theirs.numel()),
            "index_flips_vs_oracle": int(flips.sum()), "of_which_sub_margin": sub}


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (oracle = port of the reference's PyTorch-CPU path), bounded samples
# ---------------------------------------------------------------------------------------------------------------------
def _calibrated(step, threads, budget_s, max_steps):
    """Best thread count of `threads` (one step each after a warm-up step), then the median of up to max_steps steps."""
    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()
    best, best_t = None, None
    for nt in sorted({min(ncpu, c) for c in threads}):
        if time.perf_counter() - t_start > budget_s * 0.5 and best is not None:
            break
        torch.set_num_threads(nt)
        step()
        dt = step()
        if best_t is None or dt < best_t:
            best, best_t = nt, dt
    torch.set_num_threads(best)
    times = []
    while len(times) < max_steps and (time.perf_counter() - t_start < budget_s or len(times) < 3):
        times.append(step())
    times.sort()
    return times[len(times) // 2], best, len(times), ncpu


def cpu_baseline_vqvae(clips, budget_s):
    """CPU oracle's VQ-VAE train step (fwd + bwd + Adam) on this host's cores at `clips` clips per step."""
    import seeded
    from oracle import lvt_oracle as O
    enc = {k: v.requires_grad_(True) for k, v in seeded.seeded_params(seeded.VQVAE_ENCODER_SHAPES, SEED, "enc.").items()}
    dec = {k: v.requires_grad_(True) for k, v in seeded.seeded_params(seeded.VQVAE_DECODER_SHAPES, SEED, "dec.").items()}
    state = {"st": seeded.seeded_codebook_state(SEED, scale=0.05)}
    opt = torch.optim.Adam(list(enc.values()) + list(dec.values()), 3e-4, betas=(0.9, 0.9))
    x = O.normalize(seeded.seeded_input("cpu", (clips * CLIP_FRAMES, 3, 64, 64), SEED), (0.5,) * 3, (0.5,) * 3)

    def step():
        t0 = time.perf_counter()
        losses, state["st"], _ = O.vqvae_supervised_loss(enc, dec, state["st"], x)
        sum(losses.values()).backward()
        opt.step()
        opt.zero_grad()
        return time.perf_counter() - t0

    med, cores, n, ncpu = _calibrated(step, (8, 16, 32, 64, 128), budget_s, 8)
    return {"value": clips / med, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "oracle VQ-VAE train step, %d clips = %d frames per step, median of %d steps, %d threads (best of a "
                      "8..128 thread calibration on %d logical CPUs)" % (clips, clips * CLIP_FRAMES, n, cores, ncpu)}


def cpu_baseline_dsfvt(budget_s, b=4):
    """CPU oracle's DSFVT train step (fwd + bwd + RMSprop as configs/vt/DSFVT.yaml sets it) on this host's cores."""
    import seeded
    from oracle import lvt_oracle as O
    p = {k: v.requires_grad_(True) for k, v in seeded.seeded_params(seeded.dsfvt_shapes(), SEED).items()}
    opt = torch.optim.RMSprop(list(p.values()), lr=2e-5, alpha=0.95, momentum=0.9)
    block = ((1, 16, 16),) * 8
    items = [O.prepare_slices(seeded.seeded_codes("cpu.vt%d" % i, (16, 4, 16, 16), SEED), (5 + i, 0, 0), (16, 1, 1),
                              (7, 1, 1), 1) for i in range(b)]
    ctx, sl = torch.stack([d["context"] for d in items]), torch.stack([d["slice"] for d in items])
    sidx, ign = torch.stack([d["slice_idx"] for d in items]), torch.stack([d["ignore_mask"] for d in items])

    def step():
        t0 = time.perf_counter()
        loss, _ = O.vt_supervised_loss(p, ctx, sl, sidx, ign, block, block, (16, 1, 1))
        loss.backward()
        opt.step()
        opt.zero_grad()
        return time.perf_counter() - t0

    med, cores, n, ncpu = _calibrated(step, (16, 32, 64), budget_s, 6)
    return {"value": b / med, "unit": "clips/s (one slice of one clip per sample)", "cores": cores, "kind": "port",
            "sample": "oracle DSFVT train step, %d samples per step, median of %d steps, %d threads (best of a 16..64 "
                      "thread calibration on %d logical CPUs)" % (b, n, cores, ncpu)}


def cpu_baseline(budget_s, gate_state):
    """The metric on the host cores: the oracle's VQ-VAE step at 2 and 8 clips, its DSFVT step,
    combined exactly like the GPU step (every clip one VQ-VAE step and one DSFVT step): 1 / (1/v + 1/d)."""
    v2 = cpu_baseline_vqvae(2, budget_s * 0.25)
    v8 = cpu_baseline_vqvae(8, budget_s * 0.35)
    d4 = cpu_baseline_dsfvt(budget_s * 0.4)
    vbest = max(v2["value"], v8["value"])
    out = {"value": round(1.0 / (1.0 / vbest + 1.0 / d4["value"]), 3), "unit": "clips/s",
           "cores": max(v2["cores"], v8["cores"], d4["cores"]), "kind": "port",
           "sample": "CPU oracle (PyTorch-CPU fp32 restatement of the reference path, oracle/lvt_oracle.py): VQ-VAE train "
                     "step at 2 clips (%.1f clips/s, %d threads) and 8 clips (%.1f clips/s, %d threads), DSFVT train step "
                     "at 4 samples (%.2f clips/s, %d threads); combined 1/(1/vqvae_best + 1/dsfvt), bounded samples of "
                     "<= 8 / 8 / 6 steps after a per-workload thread calibration"
                     % (v2["value"], v2["cores"], v8["value"], v8["cores"], d4["value"], d4["cores"]),
           "vqvae_2_clips": v2, "vqvae_8_clips": v8, "dsfvt_4_samples": d4}
    if gate_state is not None:
        out["parity_vs_hip"] = oracle_flips(gate_state)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# generation leg (BASELINE configs[4]); runs in a CHILD process (`--generate-only`)
# ---------------------------------------------------------------------------------------------------------------------
def generation_kv_bytes(vt_cfg, batch, n_prime, tokens=256):
    """Algorithmic K/V-cache bytes of one generation run: for every generated position i of a slice the decode attention
    reads keys 0..i of K and V (hd fp32 each) in each decoder layer -- per video (16 - n_prime) slices x layers x
    tokens*(tokens+1)/2 key rows x 2 x hd x 4 B.  (N_HEAD_D is a per-layer TUPLE in the config: round 2's version of this
    line multiplied the tuple itself, which asks Python for a 4.5e12-element tuple -- the 36 TB allocation that took down
    five GPU boxes; tests/test_host_logic.py pins the arithmetic.)"""
    layers = len(vt_cfg.BLOCKS_D)
    hd = int(vt_cfg.N_HEAD_D[0]) * int(vt_cfg.DA)
    rows = tokens * (tokens + 1) // 2
    return int(batch) * (16 - int(n_prime)) * layers * rows * 2 * hd * 4


def smallm_gemm_roofline(vt):
    """The second kernel of the decode steps (45 % of the steady-state window): lvt_gemm_smallm_mfma_kernel<2>, one token per video
    through the decoder's weights.  Its algorithmic traffic is the WEIGHTS (every launch streams its matrix once, the 256 activation
    rows are a few hundred KB): bytes per decode step from the model, launches per step and the average launch time from the
    kernel trace of the steady replay (profiles/r06_generation_kernel_mix.txt -- the launches run inside a replayed hipGraph, where
    event timing is not available)."""
    import re
    dec = vt.model.decoder
    wbytes = 0
    for layer in dec.block_local_attention:
        m, f = layer.mha, layer.ffn
        wbytes += 4 * (3 * m.w_q.numel() + m.proj.weight.numel() + f[1].weight.numel() + f[3].weight.numel())
    cp = vt.model.ch_predictor
    wbytes += 4 * sum(u.weight.shape[0] * u.weight.shape[0] for u in cp.U)            # the dense d x d part; the one-hot columns are gathers
    wbytes += 4 * sum(cp._P(k).weight.numel() for k in range(cp.nc))
    try:
        txt = open(os.path.join(ROOT, "profiles", "r06_generation_kernel_mix.txt")).read()
        m1 = re.search(r"lvt_gemm_smallm_mfma_kernel<2>\s+(\d+)\s+([\d.]+) us total\s+([\d.]+) us avg", txt)
        m2 = re.search(r"lvt_decode_commit_kernel\s+(\d+)", txt)
        launches, total_us, avg_us, steps = int(m1.group(1)), float(m1.group(2)), float(m1.group(3)), int(m2.group(1))
    except Exception:
        return {"weight_bytes_per_decode_step": wbytes, "note": "profiles/r06_generation_kernel_mix.txt not found"}
    per_launch = wbytes * steps / launches
    gbs = per_launch / avg_us / 1e3
    return {"kernel": "lvt_gemm_smallm_mfma_kernel<2>", "launches_per_decode_step": round(launches / steps, 1), "avg_us": avg_us,
            "weight_bytes_per_decode_step": wbytes, "weight_bytes_per_launch": int(per_launch),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
            "note": "weights streamed per launch / average launch time of the steady replay (three decode groups on three streams: "
                    "launch times overlap).  %.0f MB of weights per step stay in the 256 MB last-level cache from step to step, "
                    "so HBM is not what binds these launches: 256 rows x N columns are 4 x N / 32 workgroups of one wave "
                    "quartet each (16 .. 128 workgroups on 256 CUs) and a launch is its latency chain -- a fused per-layer decode "
                    "kernel is the lever, not bandwidth" % (wbytes / 1e6)}


def bench_generate(device, batch):
    """End-to-end generation: VQ encode of 5 priming frames, DSFVT autoregressive sampling of the remaining 11 frames
    (incremental K/V-cache decode: 2816 single-token steps, one replayed hipGraph per decode group, position held in a
    device-side cursor), VQ decode of all 16 frames -- for `batch` videos at once.  frames/s = 16 * batch / wall time."""
    from lvt_amd.config import get_cfg
    from lvt_amd.modeling import build_model
    cfgs = []
    for path in ("configs/vt/DSFVT.yaml", "configs/vqvae/PR-DVQVAE2.yaml"):
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(ROOT, path))
        cfg.MODEL.DEVICE = device
        cfg.OUTPUT_DIR = "/tmp/lvt_bench_out"
        cfgs.append(cfg)
    cfgs[0].TEST.EVALUATORS = "VTSampler"
    torch.manual_seed(SEED)
    vt, vqvae = build_model(cfgs[0]).eval(), build_model(cfgs[1]).eval()
    n_prime = cfgs[0].TEST.VT_SAMPLER.N_PRIME
    frames = torch.rand(batch, n_prime, 3, 64, 64, generator=torch.Generator().manual_seed(5)).to(device)

    def run():
        with torch.no_grad():
            out = vqvae([{"image_sequence": frames[i]} for i in range(batch)], mode="inference")
            lat = torch.stack([o["latent"] for o in out])                       # (B, 5, 4, 16, 16)
            video = lat.new_zeros(batch, 16, lat.shape[2], 16, 16)
            video[:, :n_prime] = lat
            torch.cuda.synchronize()
            sample = vt.sample_video(video.transpose(1, 2).contiguous(), n_prime=n_prime)     # (B, 4, 16, 16, 16)
            torch.cuda.synchronize()
            return vqvae.decode(sample.transpose(1, 2).reshape(batch * 16, -1, 16, 16))
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert tuple(rec.shape) == (batch * 16, 3, 64, 64) and bool(torch.isfinite(rec).all())
    # roofline of the phase that dominates: the single-token decode steps are bound by the K/V-cache reads of the decode attention
    kv_bytes = generation_kv_bytes(cfgs[0].MODEL.AUTOREGRESSIVE.VT, batch, n_prime)
    try:
        with open(os.path.join(ROOT, "profiles", "r05_decode_attn_pmc_hbm_traffic.json")) as f:
            dec = json.load(f)
    except Exception:
        dec = {}
    import lvt_amd.modeling.meta_arch.vt as vtmod
    ngroups = (batch + vtmod.DECODE_GROUP_ROWS - 1) // vtmod.DECODE_GROUP_ROWS
    smallm = smallm_gemm_roofline(vt)
    return {"smallm_gemm": smallm, "frames_per_s": round(16 * batch / dt, 2), "videos_per_s": round(batch / dt, 3), "batch_videos": batch,
            "seconds": round(dt, 3), "decoder_steps": 11 * 256, "decode_groups": ngroups,
            "group_streams": bool(vtmod.DECODE_GROUP_STREAMS) and ngroups > 1,
            "roofline": {"bound": "hbm", "achieved": round(kv_bytes / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(kv_bytes / dt / (HBM_PEAK_GBS * 1e9), 4), "traffic": dec.get("hbm_bytes_per_launch"),
                         "traffic_unit": "HBM bytes per lvt_attn_decode_kernel launch at B = 256, all 256 keys (rocprofv3 --pmc FETCH_SIZE "
                                         "(x2) + WRITE_SIZE, profiles/r05_decode_attn_pmc_hbm_traffic.json; algorithmic K/V bytes of that "
                                         "launch: %s): eager launches -- rocprofv3 --pmc crashes on the hipGraph replay of the generation run"
                                         % dec.get("algorithmic_kv_bytes_per_launch"),
                         "kernel_alone": {"achieved": round(dec["algorithmic_kv_bytes_per_launch"] / dec["avg_us_under_pmc"] / 1e3, 1),
                                          "unit": "GB/s", "frac": round(dec["algorithmic_kv_bytes_per_launch"] / dec["avg_us_under_pmc"] / 1e3 / HBM_PEAK_GBS, 4)}
                         if dec else None,
                         "kernel": "lvt_attn_decode_kernel (K/V-cache reads of the single-token decode attention); "
                                   "`achieved` = algorithmic K/V bytes of the whole run / END-TO-END wall time (encode, "
                                   "%d decode steps of ~110 launches each replayed as one hipGraph, decode of 16 frames); "
                                   "kernel mix of the steady replay: profiles/r06_generation_kernel_mix.txt"
                                   % (11 * 256)},
            "note": "5 priming + 11 generated frames per video; random-init weights; sampling is sequential (2816 "
                    "single-token decoder steps per group of <= 256 videos); videos are replicas across GPUs"}


def cpu_baseline_generate(budget_s):
    """The reference's sampling schedule on the host cores: one FULL decoder pass per generated pixel (vt.py:121-131),
    one encoder pass per slice.  A bounded sample is timed (a few decoder passes and one encoder pass of the CPU oracle,
    one video) and extrapolated: seconds per video = 11 x (t_encoder + 256 x t_decoder_pass)."""
    import seeded
    from oracle import lvt_oracle as O
    p = seeded.seeded_params(seeded.dsfvt_shapes(), SEED)
    block = ((1, 16, 16),) * 8
    d = O.prepare_slices(seeded.seeded_codes("cpu.gen", (16, 4, 16, 16), SEED), (7, 0, 0), (16, 1, 1), (7, 1, 1), 5)
    ctx, sl, sidx = d["context"][None], d["slice"][None], d["slice_idx"][None]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    with torch.no_grad():
        O.vt_encoder(p, ctx, sidx, block, (16, 1, 1))            # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        zl = O.vt_encoder(p, ctx, sidx, block, (16, 1, 1))
        t_enc = time.perf_counter() - t0
        times = []
        t_start = time.perf_counter()
        while len(times) < 8 and (time.perf_counter() - t_start < budget_s or len(times) < 3):
            t0 = time.perf_counter()
            yl = O.vt_decoder(p, sl, zl, block)
            O.channel_predictor_pixel_probs(p, yl, (0, 3, 5), torch.full((1, 4), 0.5))
            times.append(time.perf_counter() - t0)
    times.sort()
    t_dec = times[len(times) // 2]
    per_video = 11 * (t_enc + 256 * t_dec)
    return {"value": 16.0 / per_video, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle (PyTorch-CPU restatement of the reference's schedule): 1 encoder pass (%.2f s) and the median "
                      "of %d full decoder + channel-predictor passes (%.3f s) for one video, extrapolated to "
                      "11 x (1 + 256 passes) = %.0f s per 16-frame video" % (t_enc, len(times), t_dec, per_video)}


def generate_in_child(args):
    """The generation leg in a fresh process with a hard time limit, so that the headline line is printed whatever happens
    to it.  (The five GPU boxes lost in rounds 2 and 3 by runs that included this leg were lost to HOST memory: see
    generation_kv_bytes and DESIGN.md section 4b.)"""
    cmd = [sys.executable, os.path.abspath(__file__), "--generate-only", "--generate-batch", str(args.generate_batch)]
    if args.generate_cpu_baseline:
        cmd += ["--generate-cpu-baseline", "--cpu-seconds", str(args.cpu_seconds)]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.generate_timeout)
    except subprocess.TimeoutExpired:
        return {"error": "generation child exceeded %.0f s and was killed" % args.generate_timeout}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "generation child failed (rc=%d): %s" % (r.returncode, r.stderr[-400:])}
    return json.loads(lines[-1])


# ---------------------------------------------------------------------------------------------------------------------
def run(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("LVT_BENCH_DRYRUN") == "1":
        # launch-path check without GPUs (tests/test_host_logic.py): every rank joins a gloo group and meets at a barrier
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dry_run": True, "world": dist.get_world_size(), "rank_sum": float(t)}), flush=True)
        dist.destroy_process_group()
        return
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (launch one rank per GPU, or leave WORLD_SIZE unset and "
                         "let `python bench.py --gpus N` start its own ranks)" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    if args.generate_only:
        out = bench_generate(device, args.generate_batch)
        if args.generate_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_generate(args.cpu_seconds * 0.5)
        print(json.dumps(out), flush=True)
        return
    single = bool(args.dp_single_rank) and world == 1
    if single:
        os.environ["LVT_DP_SINGLE_RANK"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    if world > 1 or single:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rccl_channels > 0:
            os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_channels)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    from lvt_amd.hip import binding as L
    vq = VqvaeLeg(device, world, rank, local_rank, args.batch_clips, args.batches, dp=world > 1 or single)
    ds = DsfvtLeg(device, world, rank, local_rank, args.dsfvt_batch, args.batches, dp=world > 1 or single)
    vq_per_step = max(1, args.dsfvt_batch // args.batch_clips)        # VQ-VAE train steps per DSFVT train step (2)
    clips_per_step = args.dsfvt_batch                                 # every one of them passes through both models

    def step(i):
        for j in range(vq_per_step):
            losses = vq.step(i * vq_per_step + j)
        return losses, ds.step(i)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # a full (generation-2) Python GC pass with torch loaded takes 60-70 ms of host time; right after a barrier the
    # host has no lead over the GPU, so one such pause would stall the device for several steps' worth of launches
    gc.collect()
    gc.disable()

    # Pass 1 -- the timed region: exactly K steps, nothing but the product path between the two barriers.
    elapsed, stats, (losses, ds_loss) = timed_steps(step, args.steps, args.warmup, world, device)
    # Pass 2 -- the same K steps again with a HIP event pair around every engine launch (on the launch stream).
    timer, instrumented_ms = instrumented(step, args.steps, args.warmup + args.steps)
    math_mode = L.get_math_mode()
    es = engine_summary(timer, args.steps, math_mode)

    # communication: what RCCL sees, and what the step costs with the reducers switched off (gradients stay local)
    reducers = list(getattr(vq.model, "_reducers", [])) + list(getattr(ds.model, "_reducers", []))
    grouped = world > 1 or single
    if grouped:     # what RCCL itself reports after a collective of this process group has run
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    comm = {"world_size": dist.get_world_size() if grouped else 1, "backend": dist.get_backend() if grouped else None,
            "ranks_seen_by_allreduce": int(probe.item()) if grouped else 1, "single_rank_group": single,
            "allreduce_bytes_per_step": int(sum(r.bytes_per_backward for r in reducers if r in vq.model._reducers) * vq_per_step
                                            + sum(r.bytes_per_backward for r in reducers if r in ds.model._reducers))
            if reducers else 0,
            "ema_allreduce_bytes_per_step": (4 * 512 * (64 + 1) * 4) * vq_per_step if world > 1 else 0,
            # CU footprint of the collectives: each RCCL channel is a workgroup competing with the one-workgroup-per-CU engine
            # launches of the backward pass; A/B it with --rccl-channels on the 8-GPU node (unmeasured here: one GPU per box)
            "rccl_channels": {"requested": args.rccl_channels or None,
                              "NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"),
                              "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS")},
            "bucket_order": ("arrival order measured in the first backward" if reducers and all(r._calibrated for r in reducers)
                             else "reverse registration order") if reducers else None}
    if grouped:
        for r in reducers:
            r.enabled = False
        n3 = max(5, args.steps // 2)
        e3, _, _ = timed_steps(step, n3, 0, world, device)
        for r in reducers:
            r.enabled = True
        comm["ms_per_step_without_grad_allreduce"] = round(e3 / n3 * 1e3, 3)
        comm["comm_exposed_ms_per_step"] = round(elapsed / args.steps * 1e3 - e3 / n3 * 1e3, 3)
        comm["note"] = ("gradient buckets are all-reduced (ReduceOp.AVG) on a side stream as their last gradient arrives; "
                        "`comm_exposed` = timed step - the same step with the reducers off (replicas then diverge: timing only)")

    extra = {}
    if not args.no_legs:
        nl = max(10, args.steps)
        extra["vqvae"] = leg_alone("vqvae", vq, nl, 3, world, device, "clips", "r06_vqvae_pmc_hbm_traffic.json",
                                   "lvt_conv_patch_kernel<0,1,2> / lvt_conv_wgrad_frames_kernel<0,1> (frame-resident 3x3 and "
                                   "4x4/stride-2 layers) + lvt_gemm_kernel<*> (1x1 and image-side layers)",
                                   not args.no_strict_f32)
        extra["dsfvt"] = leg_alone("dsfvt", ds, max(10, args.steps // 2), 2, world, device, "samples",
                                   "r06_dsfvt_pmc_hbm_traffic.json",
                                   "lvt_gemm_wide_kernel<*> (QKV / proj / FFN products, their data and weight gradients; f16x2) + "
                                   "lvt_attn_fwd_flash_kernel / lvt_attn_bwd_flash_a / _b (flash attention on fp32 operands, per-row f16x2)", not args.no_strict_f32)
        extra["host_fed"] = host_fed(vq, ds, max(6, args.steps // 2), world, device)
        v1, v2 = extra["vqvae"]["clips_per_s"], extra["dsfvt"]["samples_per_s"]
        extra["legs_combined_harmonic"] = {"clips_per_s": round(1.0 / (1.0 / v1 + 1.0 / v2), 3),
                                           "note": "1/(1/vqvae + 1/dsfvt) of the two legs timed alone: cross-check of `value`"}
    # the same job with every product on the plain fp32 MFMA instruction (LVT_MATH=f32), from the two legs timed alone in that mode
    strict_f32 = None
    if "vqvae" in extra and "strict_f32_mfma" in extra["vqvae"] and "strict_f32_mfma" in extra["dsfvt"]:
        f1, f2 = extra["vqvae"]["strict_f32_mfma"]["clips_per_s"], extra["dsfvt"]["strict_f32_mfma"]["samples_per_s"]
        strict_f32 = {"clips_per_s": round(1.0 / (1.0 / f1 + 1.0 / f2), 3), "ratio_to_value": None,
                      "note": "LVT_MATH=f32 (v_mfma_f32_32x32x2_f32 everywhere): 1/(1/vqvae + 1/dsfvt) of the two legs timed alone "
                              "in that mode (extra.*.strict_f32_mfma); `value` is measured in `math_mode`"}
    gate_state = None
    parity = None
    if not args.no_parity and rank == 0:
        parity, vq_enc, gate_state = vq_gates(vq, device, 8)
        extra["vq_encode"] = vq_enc
        parity["dsfvt_loss"] = round(float(ds_loss.detach()), 6)
        parity["vqvae_loss"] = {k: round(float(v.detach()), 6) for k, v in losses.items()}
    if grouped:
        dist.barrier()

    if rank == 0:
        del vq, ds
        gc.enable()
        gc.collect()
        torch.cuda.empty_cache()
        if not args.no_generate and world == 1:
            extra["generate"] = generate_in_child(args)
        ms = elapsed / args.steps * 1e3
        value = clips_per_step * world * args.steps / elapsed
        if strict_f32 is not None:
            strict_f32["ratio_to_value"] = round(value / strict_f32["clips_per_s"], 3)
        out = {
            "metric": "video-clips/sec/node (VQVAE+DSFVT train step, BAIR 64x64x16)",
            "value": round(value, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "math_mode": math_mode, "strict_f32": strict_f32, "data": "synthetic",
            "math": MATH_NOTE[math_mode],
            "step_ms": stats,
            "config": {"workload": "VQVAE+DSFVT train step: %d clips per GPU per step through both models -- %d x PR-DVQVAE2 "
                                   "train step (fwd+bwd+Adam, %d clips x 16 frames x 3x64x64, 4x512 EMA codebooks) + 1 x DSFVT "
                                   "train step (fwd+bwd+RMSprop, %d slices of 256 tokens x 4 code channels, 49.87M parameters); "
                                   "%d distinct batches rotated; random-init weights"
                                   % (clips_per_step, vq_per_step, args.batch_clips, args.dsfvt_batch, args.batches),
                       "global_batch_clips": clips_per_step * world, "parallelism": "dp%d" % world},
            "roofline": roofline_block(
                es, math_mode, "r06_combined_pmc_hbm_traffic.json",
                "the matrix-core engine launches of the step: lvt_gemm_wide_kernel<*>, lvt_gemm_kernel<*>, lvt_conv_patch_kernel<*>, "
                "lvt_conv_wgrad_frames_kernel<*>, lvt_attn_fwd/bwd kernels; %d launches per step, event-timed in a second "
                "pass of the same %d steps: %.2f ms of engine time in a %.2f ms instrumented step (unperturbed: %.2f ms)"
                % (es["launches_per_step"], args.steps, es["ms_per_step"], instrumented_ms, ms)),
            "comm": comm,
        }
        if parity is not None:
            out["parity"] = parity
        out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, gate_state)
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, argv)
    else:
        run(args)


if __name__ == "__main__":
    main()
