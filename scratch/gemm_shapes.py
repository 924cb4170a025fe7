"""Log the distinct tile-engine GEMM shapes of one DSFVT train step, then time each shape alone.
usage: python scratch/gemm_shapes.py            (LVT_HIP_LIB=... selects a variant library)

STALE since round 4: the batched launches now address their batches by stride from ONE base pointer (the q / k / v slabs of the
packed projection output, the two operands of a paired weight gradient), so `torch.randn_like(A)` below allocates less than the
launch reads and the replay faults.  Kept for the round-3 table in profiles/r03_gemm_shape_and_power_probes.txt; per-kind timings
of the current code are in the bench line (`roofline.per_kind`) and profiles/r05_dsfvt_kernel_stats.txt."""
raise SystemExit(__doc__)
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from lvt_amd.hip import gemm as G

dev = "cuda:0"
torch.cuda.set_device(0)
leg = bench.DsfvtLeg(dev, 1, 0, 0, 64, 2)
for i in range(2):
    leg.step(i)
torch.cuda.synchronize()

seen = collections.OrderedDict()
orig = G.gemm


def logged(A, B, C_out, M, N, K, ta=0, tb=0, **kw):
    key = (M, N, K, ta, tb, kw.get("batch_outer", 1) * kw.get("batch_inner", 1), kw.get("splits", 1), kw.get("flags", 0))
    if key not in seen:
        seen[key] = [0, (A, B, C_out, dict(kw))]
    seen[key][0] += 1
    return orig(A, B, C_out, M, N, K, ta=ta, tb=tb, **kw)


G.gemm = logged
import lvt_amd.modeling.autoregressive.vt_attention as va
leg.step(2)
torch.cuda.synchronize()
G.gemm = orig


def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tot = 0.0
print("%7s %5s %6s ta tb batch splits flags | calls |   us    TF | ms/step" % ("M", "N", "K"))
for key, (calls, (A, B, C_out, kw)) in seen.items():
    M, N, K, ta, tb, batch, splits, flags = key
    A2, B2, C2 = torch.randn_like(A) if A.dtype == torch.float32 else A, torch.randn_like(B), torch.empty_like(C_out)
    kw2 = dict(kw)
    for nm in ("res", "mask"):
        if kw2.get(nm) is not None:
            kw2[nm] = torch.randn_like(kw2[nm])
    if kw2.get("a_colsum") is not None:
        kw2["a_colsum"] = torch.empty_like(kw2["a_colsum"])
    t = timeit(lambda: orig(A2, B2, C2, M, N, K, ta=ta, tb=tb, **kw2))
    fl = 2.0 * M * N * K * batch
    tot += t * calls
    print("%7d %5d %6d %2d %2d %5d %6d %5d | %5d | %6.1f %5.1f | %6.3f" % (M, N, K, ta, tb, batch, splits, flags, calls, t * 1e3,
                                                                          fl / t / 1e9, t * calls))
print("sum ms/step: %.2f" % tot)
