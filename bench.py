#!/usr/bin/env python
"""Headline benchmark: PR-DVQVAE2 train step on synthetic BAIR-shaped clips (64x64x16), batch 32 clips
per GPU, fp32, on N MI355X of one node (BASELINE.json `configs[1]`, metric video-clips/s/node).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + backward + Adam step of the whole VQ-VAE (encoder, 4x512 EMA codebooks, decoder) on
one of `--batches` synthetic batches that are already resident in HBM (rotated step by step).  One process per GPU;
gradients are averaged with a bucketed RCCL all-reduce overlapped with backward (it joins itself before
`optimizer.step()`), EMA statistics with one fused all-reduce.  Weak scaling.
Rank 0 prints ONE JSON line: `value` / `ms_per_step` from exactly K steps between two barriers (+ per-step median / p10 / p90
from one HIP event per step); `roofline` describes the dominant kernels (the matrix-core conv / GEMM engine: implicit-GEMM
and frame-resident kernels), event-timed per launch on the launch stream in a SECOND pass of the same K steps (a timing
event is a barrier packet: inside the timed region it would cost ~15 % of the step); `cpu_baseline` is the CPU oracle (a
port of the reference's PyTorch-CPU path) timed on the host cores of the same box.  `extra` holds the same for the DSFVT
train step (own roofline block and CPU baseline), the LVT_MATH=f32 variants, the combined VQVAE+DSFVT clips/s and -- with
`--generate` -- the end-to-end generation figure.
"""
import argparse
import gc
import json
import os
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


def engine_peak(mode):
    """Ceiling of the engine in ALGORITHMIC fp32 FLOP/s: the fp32 MFMA peak in 'f32' mode; in 'bf16x3' mode every
    algorithmic product is six bf16 MFMA products (a1b1 a1b2 a2b1 a1b3 a3b1 a2b2), so the ceiling is bf16 peak / 6."""
    return FP32_MFMA_PEAK_TFLOPS if mode == "f32" else BF16_MFMA_PEAK_TFLOPS / 6.0
CLIP_FRAMES = 16


MATH_NOTE = {
    "f32": "fp32 in, fp32 out, v_mfma_f32_32x32x2_f32 products, fp32 accumulation",
    "bf16x3": "fp32 in, fp32 out, fp32 accumulation; every fp32 product formed exactly from a 3-way bf16 split of both "
              "operands (six v_mfma_f32_32x32x16_bf16 per block, dropped terms < 2^-24 |a||b|); measured error against "
              "fp64 is not larger than the plain fp32 MFMA path's (profiles/r01_math_mode_accuracy.txt, "
              "tests/test_gpu_engine.py::test_math_modes_accuracy); LVT_MATH=f32 selects the plain fp32 instruction",
}
PEAK_NOTE = {
    "f32": "dense fp32 MFMA peak (MI355X_MICROARCH.md)",
    "bf16x3": "dense bf16 MFMA peak 2516.6 TFLOP/s / 6 MFMA products per algorithmic fp32 product; `achieved` counts "
              "ALGORITHMIC fp32 FLOPs only (not the 6x executed bf16 FLOPs); an MFMA-only loop on random bf16 operands "
              "sustains 71 % of that peak on this part (power throttling by operand toggling, "
              "profiles/r01_ubench_engine_bounds.txt), i.e. ~300 TFLOP/s in these units",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batches", type=int, default=4, help="distinct synthetic batches rotated through the steps")
    ap.add_argument("--batch-clips", type=int, default=32, help="clips per GPU per step (BASELINE: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dsfvt", action="store_true", help="skip the secondary DSFVT train-step figure")
    ap.add_argument("--dsfvt-batch", type=int, default=64)
    ap.add_argument("--generate", action="store_true",
                    help="also run the secondary end-to-end generation figure (BASELINE configs[4]).  Opt-in: three of three "
                         "runs of this leg lost their GPU box late in round 2 (all four runs without it in the same window "
                         "were fine; the same code had run it six times that day), so the default run, which has to deliver "
                         "the headline, does not risk it; the last measured line is profiles/r02_bench_final.json")
    ap.add_argument("--no-generate", action="store_true", help="(accepted for compatibility; generation is off by default)")
    ap.add_argument("--generate-batch", type=int, default=768, help="videos generated at once (decoded as groups of <= 256 on separate streams)")
    ap.add_argument("--no-strict-f32", action="store_true", help="skip the secondary LVT_MATH=f32 figure")
    ap.add_argument("--generate-cpu-baseline", action="store_true",
                    help="also time the reference's sampling schedule on the host cores (bounded sample, extrapolated)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample")
    return ap.parse_args()


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
    18+ ms apart, so the barrier packet an event inserts is not measurable)."""
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


def traffic_from_profile(name):
    """HBM bytes per engine launch from this round's committed PMC passes (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
    separate runs of this command, summarised by scratch/pmc_summary.py): counters cannot be read inside the run."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return round(json.load(f)["hbm_bytes_per_launch"])
    except Exception:
        return None


def build_vqvae(device, seed):
    from lvt_amd.config import get_cfg
    from lvt_amd.modeling import build_model
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs/vqvae/PR-DVQVAE2.yaml"))
    cfg.MODEL.DEVICE = device
    cfg.OUTPUT_DIR = "/tmp/lvt_bench_out"
    torch.manual_seed(seed)
    model = build_model(cfg)
    model.train()
    return cfg, model


def vqvae_step(model, optimizers, data, storage_iter):
    from lvt_amd.utils.events import EventStorage
    with EventStorage(storage_iter):
        losses = model(data, mode="supervised")
    total = sum(losses.values())
    total.backward()
    for o in optimizers:          # (under data parallelism the gradient all-reduce joins itself before step)
        o["optimizer"].step()
    for o in optimizers:
        o["optimizer"].zero_grad()
    return losses


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
    return {"achieved": achieved, "ms_per_step": tot_ms / steps, "launches_per_step": launches // steps,
            "flops_per_launch": tot_fl / max(launches, 1), "per_kind": per_kind, "frac": achieved / engine_peak(mode)}


def bench_dsfvt(device, world, rank, steps, warmup, batch, nbatches, strict_f32=True, cpu_seconds=0.0):
    """Second workload of the metric: DSFVT train step (fwd + bwd + RMSprop) on synthetic code clips, one random
    subscale slice per clip (BASELINE.json configs[2] / [3]); reported under `extra.dsfvt` with its own roofline block."""
    from lvt_amd.config import get_cfg
    from lvt_amd.data.dataset_mapper import prepare_slices_batch
    from lvt_amd.hip import binding as L
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs/vt/DSFVT.yaml"))
    cfg.MODEL.DEVICE = device
    cfg.OUTPUT_DIR = "/tmp/lvt_bench_out"
    torch.manual_seed(29871897 + rank)
    model = build_model(cfg)
    model.train()
    optimizers, _ = model.configure_optimizers_and_checkpointers()
    if world > 1:
        model.wrap_parallel(device_ids=[0], broadcast_buffers=False)
    v = cfg.MODEL.AUTOREGRESSIVE.VT
    g = torch.Generator(device="cpu").manual_seed(4321 + rank)
    batches = []
    for _ in range(nbatches):                      # distinct clips and slice offsets per batch, built on the device
        codes = torch.randint(0, v.NV, (batch, 16, v.NC, 16, 16), generator=g).to(device)
        abcs = [(int(a), 0, 0) for a in torch.randint(v.N_PRIME, 16, (batch,), generator=g)]
        batches.append(prepare_slices_batch(codes, abcs, v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE))

    def step(i):
        ctx, sl, sidx, ign = batches[i % nbatches]
        with EventStorage(i):
            loss = model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
        loss.backward()                            # (the gradient all-reduce joins itself before optimizer.step)
        for o in optimizers:
            o["optimizer"].step()
        for o in optimizers:
            o["optimizer"].zero_grad()
        return loss

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    gc.collect()
    elapsed, stats, loss = timed_steps(step, steps, warmup, world, device)
    L.TIMER = L.KernelTimer()          # instrumented pass (per-launch events), see main()
    for i in range(steps):
        step(warmup + steps + i)
    torch.cuda.synchronize()
    timer, L.TIMER = L.TIMER, None
    mode = L.get_math_mode()
    es = engine_summary(timer, steps, mode)
    out = {"samples_per_s": round(batch * world * steps / elapsed, 2), "ms_per_step": round(elapsed / steps * 1e3, 2),
           "step_ms": stats, "steps": steps, "warmup": warmup,
           "batch_per_gpu": batch, "loss": round(float(loss.detach()), 5),
           "engine_tflops": round(es["achieved"], 2), "engine_ms_per_step": round(es["ms_per_step"], 2),
           "engine_frac_of_peak": round(es["frac"], 4),
           "roofline": {"bound": "mfma", "achieved": round(es["achieved"], 2), "peak": round(engine_peak(mode), 1),
                        "unit": "TFLOP/s", "frac": round(es["frac"], 4),
                        "traffic": traffic_from_profile("r02_dsfvt_pmc_hbm_traffic.json"),
                        "traffic_unit": "HBM bytes per engine launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate "
                                        "passes, profiles/r02_dsfvt_pmc_hbm_traffic.txt); algorithmic flops per launch = %.3e"
                                        % es["flops_per_launch"],
                        "kernel": "lvt_gemm_kernel<*> (LN'd tokens x packed QKV / proj / FFN weights, their data and weight "
                                  "gradients, attention-backward GEMMs) + lvt_attn_fwd_kernel; %d launches per step, "
                                  "event-timed in a second pass of the same %d steps: %.2f ms of engine time"
                                  % (es["launches_per_step"], steps, es["ms_per_step"]),
                        "per_kind": es["per_kind"]},
           "note": "one subscale slice (256 tokens x 4 code channels) of one 16-frame clip per sample, %d distinct batches "
                   "rotated; fp32 data; 49.87M parameters; `achieved` counts algorithmic fp32 FLOPs of the engine and "
                   "fused-attention launches (one-hot products are gathers and are not counted)" % nbatches}
    if L.get_math_mode() != "f32" and strict_f32:
        # MFMA utilisation of the attention / MLP GEMMs on the plain fp32 instruction
        L.set_math_mode("f32")
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        L.TIMER = L.KernelTimer()
        n2 = max(3, steps // 3)
        for i in range(n2):
            step(i)
        torch.cuda.synchronize()
        timer, L.TIMER = L.TIMER, None
        L.set_math_mode(mode)
        e2 = engine_summary(timer, n2, "f32")
        out["strict_f32_mfma"] = {"engine_tflops": round(e2["achieved"], 2), "mfma_utilisation": round(e2["frac"], 4),
                                  "note": "LVT_MATH=f32: the same launches on v_mfma_f32_32x32x2_f32, fraction of "
                                          "the 157.3 TFLOP/s fp32 MFMA peak"}
    if rank == 0 and world == 1 and cpu_seconds > 0:
        del model, optimizers, batches
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline_dsfvt(cpu_seconds)
    return out


def bench_generate(device, batch):
    """Secondary figure (BASELINE.json configs[4]): end-to-end generation -- VQ encode of 5 priming frames,
    DSFVT autoregressive sampling of the remaining 11 frames (incremental K/V-cache decode), VQ decode of all
    16 frames -- for `batch` videos at once on one GPU.  frames/s = 16 * batch / wall time."""
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
    torch.manual_seed(29871897)
    vt, vqvae = build_model(cfgs[0]).eval(), build_model(cfgs[1]).eval()
    n_prime = cfgs[0].TEST.VT_SAMPLER.N_PRIME
    frames = torch.rand(batch, n_prime, 3, 64, 64, generator=torch.Generator().manual_seed(5)).to(device)

    def run():
        with torch.no_grad():
            out = vqvae([{"image_sequence": frames[i]} for i in range(batch)], mode="inference")
            lat = torch.stack([o["latent"] for o in out])                       # (B, 5, 4, 16, 16)
            video = lat.new_zeros(batch, 16, lat.shape[2], 16, 16)
            video[:, :n_prime] = lat
            sample = vt.sample_video(video.transpose(1, 2).contiguous(), n_prime=n_prime)     # (B, 4, 16, 16, 16)
            rec = vqvae.decode(sample.transpose(1, 2).reshape(batch * 16, -1, 16, 16))
            return rec
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # roofline of the phase that dominates (the single-token decode steps are bound by the K/V-cache reads of the decode
    # attention): algorithmic bytes = for every generated position i of a slice, keys 0..i of K and V (hd fp32 each) in each
    # of the decoder layers -- per video 11 slices x 8 layers x (256 * 257 / 2) key rows x 2 x 4 KiB
    v = cfgs[0].MODEL.AUTOREGRESSIVE.VT
    hd = v.N_HEAD_D * v.DA
    kv_bytes = batch * (16 - n_prime) * len(v.BLOCKS_D) * (256 * 257 // 2) * 2 * hd * 4
    return {"frames_per_s": round(16 * batch / dt, 2), "videos_per_s": round(batch / dt, 3), "batch_videos": batch,
            "seconds": round(dt, 3), "decoder_steps": 11 * 256,
            "roofline": {"bound": "hbm", "achieved": round(kv_bytes / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(kv_bytes / dt / 8e12, 4), "traffic": None,
                         "kernel": "lvt_attn_decode_kernel (K/V-cache reads of the single-token decode attention); "
                                   "`achieved` = algorithmic K/V bytes of the whole run / END-TO-END wall time (encode, "
                                   "%d decode steps of ~90 launches each, decode of 16 frames); the kernel alone streams "
                                   "the caches at 6.0 TB/s (profiles/r01_generation_kernel_mix.txt)" % (11 * 256)},
            "note": "5 priming + 11 generated frames per video; random-init weights; sampling is sequential "
                    "(2816 single-token decoder steps per group of <= 256 videos; the groups of a batch run on separate "
                    "streams), videos are replicas across GPUs"}


def cpu_baseline_generate(budget_s):
    """The reference's sampling schedule on the host cores: one FULL decoder pass per generated pixel (vt.py:121-131),
    one encoder pass per slice.  A bounded sample is timed (a few decoder passes and one encoder pass of the CPU oracle,
    one video) and extrapolated: seconds per video = 11 x (t_encoder + 256 x t_decoder_pass)."""
    import seeded
    from oracle import lvt_oracle as O
    seed = 29871897
    p = seeded.seeded_params(seeded.dsfvt_shapes(), seed)
    block = ((1, 16, 16),) * 8
    d = O.prepare_slices(seeded.seeded_codes("cpu.gen", (16, 4, 16, 16), seed), (7, 0, 0), (16, 1, 1), (7, 1, 1), 5)
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


def cpu_baseline(batch_clips, budget_s):
    """Time the CPU oracle's VQ-VAE train step (fwd + bwd + Adam) on this host's cores."""
    import seeded
    from oracle import lvt_oracle as O
    seed = 29871897
    enc = {k: v.requires_grad_(True) for k, v in seeded.seeded_params(seeded.VQVAE_ENCODER_SHAPES, seed, "enc.").items()}
    dec = {k: v.requires_grad_(True) for k, v in seeded.seeded_params(seeded.VQVAE_DECODER_SHAPES, seed, "dec.").items()}
    state = {"st": seeded.seeded_codebook_state(seed, scale=0.05)}
    opt = torch.optim.Adam(list(enc.values()) + list(dec.values()), 3e-4, betas=(0.9, 0.9))
    clips = 2
    x = O.normalize(seeded.seeded_input("cpu", (clips * CLIP_FRAMES, 3, 64, 64), seed), (0.5,) * 3, (0.5,) * 3)

    def step():
        t0 = time.perf_counter()
        losses, state["st"], _ = O.vqvae_supervised_loss(enc, dec, state["st"], x)
        sum(losses.values()).backward()
        opt.step()
        opt.zero_grad()
        return time.perf_counter() - t0

    # PyTorch-CPU does not scale to every hardware thread on these small convolutions: calibrate the
    # thread count (one step each, after one warm-up step) and time the best one.
    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()
    best, best_t = None, None
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        if time.perf_counter() - t_start > budget_s * 0.5 and best is not None:
            break
        torch.set_num_threads(nt)
        step()
        dt = step()
        if best_t is None or dt < best_t:
            best, best_t = nt, dt
    cores = best
    torch.set_num_threads(cores)
    times = []
    while len(times) < 10 and (time.perf_counter() - t_start < budget_s or len(times) < 3):
        times.append(step())
    times.sort()
    med = times[len(times) // 2]
    return {"value": clips / med, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "oracle (PyTorch-CPU fp32 restatement of the reference path) VQ-VAE train step, "
                      "%d clips = %d frames per step, median of %d steps, %d threads (best of a 8..128 thread "
                      "calibration on %d logical CPUs)" % (clips, clips * CLIP_FRAMES, len(times), cores, ncpu)}


def cpu_baseline_dsfvt(budget_s):
    """CPU oracle's DSFVT train step (fwd + bwd + RMSprop as configs/vt/DSFVT.yaml sets it) on this host's cores."""
    import seeded
    from oracle import lvt_oracle as O
    seed, b = 29871897, 4
    p = {k: v.requires_grad_(True) for k, v in seeded.seeded_params(seeded.dsfvt_shapes(), seed).items()}
    opt = torch.optim.RMSprop(list(p.values()), lr=2e-5, alpha=0.95, momentum=0.9)
    block = ((1, 16, 16),) * 8
    items = [O.prepare_slices(seeded.seeded_codes("cpu.vt%d" % i, (16, 4, 16, 16), seed), (5 + i, 0, 0), (16, 1, 1),
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

    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()
    best, best_t = None, None
    for nt in sorted({min(ncpu, c) for c in (16, 32, 64)}):
        if time.perf_counter() - t_start > budget_s * 0.5 and best is not None:
            break
        torch.set_num_threads(nt)
        step()
        dt = step()
        if best_t is None or dt < best_t:
            best, best_t = nt, dt
    torch.set_num_threads(best)
    times = []
    while len(times) < 8 and (time.perf_counter() - t_start < budget_s or len(times) < 3):
        times.append(step())
    times.sort()
    med = times[len(times) // 2]
    return {"value": b / med, "unit": "samples/s (= clips/s: one slice of one clip per sample)", "cores": best, "kind": "port",
            "sample": "oracle (PyTorch-CPU fp32 restatement of the reference path) DSFVT train step, %d samples per step, "
                      "median of %d steps, %d threads (best of a 16..64 thread calibration on %d logical CPUs)"
                      % (b, len(times), best, ncpu)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    from lvt_amd.hip import binding as L
    cfg, model = build_vqvae(device, 29871897 + rank)
    optimizers, _ = model.configure_optimizers_and_checkpointers()
    if world > 1:
        model.wrap_parallel(device_ids=[local_rank], broadcast_buffers=False)

    # synthetic clips, resident in HBM before the timed region; `--batches` distinct batches are rotated
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    batches = []
    for _ in range(args.batches):
        clips = torch.rand(args.batch_clips, CLIP_FRAMES, 3, 64, 64, generator=g).to(device)
        batches.append([{"image_sequence": clips[i]} for i in range(args.batch_clips)])

    def step(i):
        return vqvae_step(model, optimizers, batches[i % args.batches], i)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # a full (generation-2) Python GC pass with torch loaded takes 60-70 ms of host time; right after a barrier the
    # host has no lead over the GPU, so one such pause would stall the device for 3 steps' worth of launches
    gc.collect()
    gc.freeze()

    # Pass 1 -- the timed region: exactly K steps, nothing but the product path between the two barriers.
    elapsed, stats, losses = timed_steps(step, args.steps, args.warmup, world, device)

    # Pass 2 -- the same K steps again with a HIP event pair around every engine launch (on the launch stream) for
    # the roofline block.  It is a separate pass because a timing event is a barrier packet: it serialises
    # consecutive launches (the next kernel can no longer fill CUs while the previous one drains), which costs
    # ~15% of the step and would be charged to `value` if both ran together.
    L.TIMER = L.KernelTimer()
    t1 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + args.steps + i)
    torch.cuda.synchronize()
    instrumented_ms = (time.perf_counter() - t1) / args.steps * 1e3
    timer, L.TIMER = L.TIMER, None
    math_mode = L.get_math_mode()

    strict = None
    if math_mode != "f32" and not args.no_strict_f32:
        # the same timed region on the plain fp32 MFMA instruction (v_mfma_f32_32x32x2_f32), for reference
        L.set_math_mode("f32")
        for i in range(2):
            step(i)
        n2 = max(5, args.steps // 2)
        e2, st2, _ = timed_steps(step, n2, 2, world, device)
        strict = {"clips_per_s": round(args.batch_clips * world * n2 / e2, 3), "ms_per_step": round(e2 / n2 * 1e3, 3),
                  "steps": n2, "note": "LVT_MATH=f32: identical step on v_mfma_f32_32x32x2_f32 (peak 157.3 TFLOP/s)"}
        L.set_math_mode(math_mode)

    extra = {}
    if not args.no_dsfvt:
        del model, optimizers, batches
        torch.cuda.empty_cache()
        extra["dsfvt"] = bench_dsfvt(device, world, rank, max(10, args.steps // 2), 3, args.dsfvt_batch, args.batches,
                                     strict_f32=not args.no_strict_f32,
                                     cpu_seconds=0.0 if args.no_cpu_baseline else args.cpu_seconds * 0.75)
    if args.generate and not args.no_generate and rank == 0 and world == 1:
        torch.cuda.empty_cache()
        extra["generate"] = bench_generate(device, args.generate_batch)
        if args.generate_cpu_baseline:
            extra["generate"]["cpu_baseline"] = cpu_baseline_generate(args.cpu_seconds * 0.5)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch_clips * world * args.steps / elapsed
        es = engine_summary(timer, args.steps, math_mode)
        out = {
            "metric": "video-clips/sec/node (VQ-VAE PR-DVQVAE2 train step, BAIR 64x64x16)",
            "value": round(value, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "math": MATH_NOTE[math_mode],
            "step_ms": stats,
            "config": {"workload": "PR-DVQVAE2 train step (fwd+bwd+Adam), %d clips x 16 frames x 3x64x64 per GPU, "
                                   "4x512 EMA codebooks; %d distinct batches rotated" % (args.batch_clips, args.batches),
                       "global_batch_clips": args.batch_clips * world, "parallelism": "dp%d" % world,
                       "loss": {k: round(float(v.detach()), 6) for k, v in losses.items()}},
            "roofline": {"bound": "mfma", "achieved": round(es["achieved"], 2), "peak": round(engine_peak(math_mode), 1),
                         "unit": "TFLOP/s", "frac": round(es["frac"], 4),
                         "traffic": traffic_from_profile("r02_vqvae_pmc_hbm_traffic.json"),
                         "traffic_unit": "HBM bytes per engine launch (rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + "
                                         "WRITE_SIZE in separate passes of this command, profiles/r02_vqvae_pmc_hbm_traffic.txt); "
                                         "algorithmic flops per launch = %.3e" % es["flops_per_launch"],
                         "kernel": "lvt_gemm_kernel<*> (implicit-GEMM engine: conv fwd / bwd-data / bwd-weight), %d "
                                   "launches per step; event-timed in a second pass of the same %d steps: %.2f ms of "
                                   "engine time in a %.2f ms instrumented step (unperturbed step: %.2f ms)"
                                   % (es["launches_per_step"], args.steps, es["ms_per_step"], instrumented_ms, ms),
                         "peak_note": PEAK_NOTE[math_mode],
                         "per_kind": es["per_kind"]},
        }
        if strict is not None:
            extra["strict_f32_mfma"] = strict
        if "dsfvt" in extra:
            # BASELINE.json words the metric as "VQVAE+DSFVT train step": a clip that takes one VQ-VAE train step AND one
            # DSFVT train step (one slice of it per step, as the reference trains) on the same GPUs, one after the other
            v1, v2 = value, extra["dsfvt"]["samples_per_s"]
            extra["combined_vqvae_dsfvt"] = {
                "clips_per_s": round(1.0 / (1.0 / v1 + 1.0 / v2), 3),
                "note": "harmonic combination 1/(1/vqvae + 1/dsfvt) of the two measured train-step rates: clips/s when "
                        "every clip gets one VQ-VAE step and one DSFVT step on the same %d GPU(s)" % world}
        out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.batch_clips, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
