#!/usr/bin/env python
"""Headline benchmark: PR-DVQVAE2 train step on synthetic BAIR-shaped clips (64x64x16), batch 32 clips
per GPU, fp32, on N MI355X of one node (BASELINE.json `configs[1]`, metric video-clips/s/node).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + backward + Adam step of the whole VQ-VAE (encoder, 4x512 EMA codebooks, decoder) on
one batch that is already resident in HBM.  One process per GPU; gradients are averaged with a bucketed
RCCL all-reduce overlapped with backward, EMA statistics with one fused all-reduce.  Weak scaling.
Rank 0 prints ONE JSON line; `roofline` describes the dominant kernel (the fp32-MFMA implicit-GEMM engine,
timed per launch with HIP events on the launch stream inside the timed region) and `cpu_baseline` is the
CPU oracle (a port of the reference's PyTorch-CPU path) timed on the host cores of the same box.
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-clips", type=int, default=32, help="clips per GPU per step (BASELINE: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dsfvt", action="store_true", help="skip the secondary DSFVT train-step figure")
    ap.add_argument("--dsfvt-batch", type=int, default=64)
    ap.add_argument("--no-generate", action="store_true", help="skip the secondary generation figure")
    ap.add_argument("--generate-batch", type=int, default=768, help="videos generated at once (decoded as groups of <= 256 on separate streams)")
    ap.add_argument("--no-strict-f32", action="store_true", help="skip the secondary LVT_MATH=f32 figure")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample")
    return ap.parse_args()


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
    for o in optimizers:
        o["optimizer"].step()
    for o in optimizers:
        o["optimizer"].zero_grad()
    return losses


def bench_dsfvt(device, world, rank, steps, warmup, batch, strict_f32=True):
    """Secondary figure: DSFVT train step (fwd + bwd + RMSprop) on synthetic code clips, one random
    subscale slice per clip (BASELINE.json configs[2]); reported under `extra.dsfvt`."""
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
    codes = torch.randint(0, v.NV, (batch, 16, v.NC, 16, 16), generator=g).to(device)
    abcs = [(int(a), 0, 0) for a in torch.randint(v.N_PRIME, 16, (batch,), generator=g)]
    ctx, sl, sidx, ign = prepare_slices_batch(codes, abcs, v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE)

    def step(i):
        with EventStorage(i):
            loss = model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
        loss.backward()
        for o in optimizers:
            o["optimizer"].step()
        for o in optimizers:
            o["optimizer"].zero_grad()
        return loss

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    gc.collect()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    L.TIMER = L.KernelTimer()          # instrumented pass (per-launch events), see main()
    for i in range(steps):
        step(warmup + steps + i)
    torch.cuda.synchronize()
    timer, L.TIMER = L.TIMER, None
    summ = timer.summary()
    eng_ms = sum(v_["ms"] for v_ in summ.values())
    eng_fl = sum(v_["flops"] for v_ in summ.values())
    out = {"samples_per_s": round(batch * world * steps / elapsed, 2), "ms_per_step": round(elapsed / steps * 1e3, 2),
           "batch_per_gpu": batch, "loss": round(float(loss.detach()), 5),
           "engine_tflops": round(eng_fl / (eng_ms * 1e-3) / 1e12, 2) if eng_ms else None,
           "engine_ms_per_step": round(eng_ms / steps, 2),
           "engine_frac_of_peak": round(eng_fl / (eng_ms * 1e-3) / 1e12 / engine_peak(L.get_math_mode()), 4) if eng_ms else None,
           "note": "one subscale slice (256 tokens x 4 code channels) of one 16-frame clip per sample; fp32 data; 49.87M "
                   "parameters; engine_tflops counts algorithmic fp32 GEMM FLOPs of the engine launches (event-timed "
                   "in a second pass)"}
    if L.get_math_mode() != "f32" and strict_f32:
        # MFMA utilisation of the attention / MLP GEMMs on the plain fp32 instruction (target: >= 50 %)
        mode = L.get_math_mode()
        L.set_math_mode("f32")
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        L.TIMER = L.KernelTimer()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        timer, L.TIMER = L.TIMER, None
        L.set_math_mode(mode)
        s2 = timer.summary()
        ms2 = sum(v_["ms"] for v_ in s2.values())
        fl2 = sum(v_["flops"] for v_ in s2.values())
        out["strict_f32_mfma"] = {"engine_tflops": round(fl2 / (ms2 * 1e-3) / 1e12, 2),
                                  "mfma_utilisation": round(fl2 / (ms2 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                                  "note": "LVT_MATH=f32: the same GEMM launches on v_mfma_f32_32x32x2_f32, fraction of "
                                          "the 157.3 TFLOP/s fp32 MFMA peak"}
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
    return {"frames_per_s": round(16 * batch / dt, 2), "videos_per_s": round(batch / dt, 3), "batch_videos": batch,
            "seconds": round(dt, 3), "decoder_steps": 11 * 256,
            "note": "5 priming + 11 generated frames per video; random-init weights; sampling is sequential "
                    "(2816 single-token decoder steps per group of <= 256 videos; the groups of a batch run on separate "
                    "streams), videos are replicas across GPUs"}


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

    # synthetic clips, resident in HBM before the timed region
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    clips = torch.rand(args.batch_clips, CLIP_FRAMES, 3, 64, 64, generator=g).to(device)
    data = [{"image_sequence": clips[i]} for i in range(args.batch_clips)]

    for i in range(args.warmup):
        vqvae_step(model, optimizers, data, i)
    torch.cuda.synchronize()
    # a full (generation-2) Python GC pass with torch loaded takes 60-70 ms of host time; right after a barrier the
    # host has no lead over the GPU, so one such pause would stall the device for 3 steps' worth of launches
    gc.collect()
    gc.freeze()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    # Pass 1 -- the timed region: exactly K steps, nothing but the product path between the two barriers.
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = vqvae_step(model, optimizers, data, args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # Pass 2 -- the same K steps again with a HIP event pair around every engine launch (on the launch stream) for
    # the roofline block.  It is a separate pass because a timing event is a barrier packet: it serialises
    # consecutive launches (the next kernel can no longer fill CUs while the previous one drains), which costs
    # ~15% of the step and would be charged to `value` if both ran together.
    L.TIMER = L.KernelTimer()
    t1 = time.perf_counter()
    for i in range(args.steps):
        vqvae_step(model, optimizers, data, args.warmup + args.steps + i)
    torch.cuda.synchronize()
    instrumented_ms = (time.perf_counter() - t1) / args.steps * 1e3
    timer, L.TIMER = L.TIMER, None
    math_mode = L.get_math_mode()

    strict = None
    if math_mode != "f32" and not args.no_strict_f32:
        # the same timed region on the plain fp32 MFMA instruction (v_mfma_f32_32x32x2_f32), for reference
        L.set_math_mode("f32")
        for i in range(2):
            vqvae_step(model, optimizers, data, i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t2 = time.perf_counter()
        for i in range(args.steps):
            vqvae_step(model, optimizers, data, i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e2 = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(e2, op=dist.ReduceOp.MAX)
        strict = {"clips_per_s": round(args.batch_clips * world * args.steps / float(e2.item()), 3),
                  "ms_per_step": round(float(e2.item()) / args.steps * 1e3, 3),
                  "note": "LVT_MATH=f32: identical step on v_mfma_f32_32x32x2_f32 (peak 157.3 TFLOP/s)"}
        L.set_math_mode(math_mode)

    extra = {}
    if not args.no_dsfvt:
        del model, optimizers, clips, data
        torch.cuda.empty_cache()
        extra["dsfvt"] = bench_dsfvt(device, world, rank, max(3, args.steps // 4), 2, args.dsfvt_batch,
                                     strict_f32=not args.no_strict_f32)
    if not args.no_generate and rank == 0 and world == 1:
        torch.cuda.empty_cache()
        extra["generate"] = bench_generate(device, args.generate_batch)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch_clips * world * args.steps / elapsed
        summ = timer.summary()
        eng = {k: v for k, v in summ.items() if k.startswith("conv_") or k.startswith("gemm_")}
        tot_ms = sum(v["ms"] for v in eng.values())
        tot_fl = sum(v["flops"] for v in eng.values())
        launches = sum(v["launches"] for v in eng.values())
        achieved = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        per_kind = {k: {"launches": v["launches"], "avg_us": round(v["ms"] / v["launches"] * 1e3, 1),
                        "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in sorted(eng.items())}
        traffic = None
        try:      # HBM bytes per engine launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate runs)
            with open(os.path.join(ROOT, "profiles", "r01_vqvae_pmc_hbm_traffic_bf16x3.json")) as f:
                traffic = round(json.load(f)["hbm_bytes_per_launch"])
        except Exception:
            pass
        out = {
            "metric": "video-clips/sec/node (VQ-VAE PR-DVQVAE2 train step, BAIR 64x64x16)",
            "value": round(value, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "math": MATH_NOTE[math_mode],
            "config": {"workload": "PR-DVQVAE2 train step (fwd+bwd+Adam), %d clips x 16 frames x 3x64x64 per GPU, "
                                   "4x512 EMA codebooks" % args.batch_clips,
                       "global_batch_clips": args.batch_clips * world, "parallelism": "dp%d" % world,
                       "loss": {k: round(float(v.detach()), 6) for k, v in losses.items()}},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(engine_peak(math_mode), 1),
                         "unit": "TFLOP/s", "frac": round(achieved / engine_peak(math_mode), 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, "
                                         "profiles/r01_vqvae_pmc_hbm_traffic_bf16x3.txt); algorithmic flops per launch = %.3e" % (tot_fl / max(launches, 1)),
                         "kernel": "lvt_gemm_kernel<*> (implicit-GEMM engine: conv fwd / bwd-data / bwd-weight), %d "
                                   "launches per step; event-timed in a second pass of the same %d steps: %.2f ms of "
                                   "engine time in a %.2f ms instrumented step (unperturbed step: %.2f ms)"
                                   % (launches // args.steps, args.steps, tot_ms / args.steps, instrumented_ms, ms),
                         "peak_note": PEAK_NOTE[math_mode],
                         "per_kind": per_kind},
        }
        if strict is not None:
            extra["strict_f32_mfma"] = strict
        out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.batch_clips, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
