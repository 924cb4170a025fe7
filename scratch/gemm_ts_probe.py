"""Phase timestamps of the tile-engine workgroups (needs the timestamp build, LVT_HIP_LIB=scratch/variants/lib_gts.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lvt_amd.hip import gemm as G
dev = "cuda:0"
M = 16384
for (N, K, tb, kind) in ((512, 512, 0, "randn"), (512, 3072, 0, "randn"), (512, 3072, 0, "zeros")):
    mk = torch.randn if kind == "randn" else torch.zeros
    A = mk(M, K, device=dev); B = mk(N, K, device=dev) if tb == 0 else mk(K, N, device=dev)
    nwg = (M // 128) * (N // 128)
    Cb = torch.empty(M * N + nwg * 16, device=dev)
    for it in range(30):                      # a sustained run: the clock settles
        G.gemm(A, B, Cb, M, N, K, tb=tb)
    torch.cuda.synchronize()
    raw = Cb[M * N:].view(torch.int64).view(nwg, 8).cpu().numpy().astype(np.float64)
    t = raw[:, :4] / 100.0
    mhz = (raw[:, 6] - raw[:, 5]) / (t[:, 2] - t[:, 1])
    print("%s operands: shader clock over the main loop (s_memtime / s_memrealtime): mean %.0f MHz (p10 %.0f, p90 %.0f)" % (kind, mhz.mean(), np.percentile(mhz, 10), np.percentile(mhz, 90)))
    t0 = t[:, 0].min()
    d = np.diff(t, axis=1)
    print("NT"[tb] + " %dx%dx%d: %d workgroups; span %.1f us; starts within %.2f us" % (M, N, K, nwg, t[:, 3].max() - t0, (t[:, 0] - t0).max()))
    for k, nm in enumerate(("prologue (tile decode, first fetch, split, store, barrier)", "main loop", "epilogue")):
        print("   %-60s mean %6.2f us  p10 %6.2f  p90 %6.2f" % (nm, d[:, k].mean(), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
    print("   end times: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(t[:, 3] - t0, [10, 50, 90, 100])))
