"""Is the tile engine clock/power-limited?  Same launches on random, constant and zero operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvt_amd.hip import gemm as G
dev = "cuda:0"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M = 16384
for (N, K) in ((512, 512), (512, 3072)):
    for kind in ("randn", "ones", "zeros", "randn"):
        mk = {"randn": torch.randn, "ones": torch.ones, "zeros": torch.zeros}[kind]
        A = mk(M, K, device=dev); B = mk(N, K, device=dev); C = torch.empty(M, N, device=dev)
        t = timeit(lambda: G.gemm(A, B, C, M, N, K))
        print("NT %dx%dx%d %-6s %7.1f us %6.1f TF" % (M, N, K, kind, t * 1e3, 2.0 * M * N * K / t / 1e9))
