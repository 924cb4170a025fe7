"""Fixed per-tile cost of the engine: GEMMs with 1, 2, 4, 16 k-tiles at 512 tiles (one wave of workgroups)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvt_amd.hip import gemm as G
dev = "cuda:0"
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M, N = 16384, 512
for K in (32, 64, 128, 256, 512, 1024):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    t = timeit(lambda: G.gemm(x, w, out, M, N, K))
    print("NT %dx%dx%-5d %6.1f us  (%d k-tiles)  %.1f TF" % (M, N, K, t, K // 32, 2 * M * N * K / t / 1e6))
