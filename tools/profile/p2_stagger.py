import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L
from lvt_amd.hip import gemm as G
dev = torch.device("cuda:0")
L.set_math_mode("f16x2")
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
def pack(x):
    dst = torch.empty_like(x); am = L.amax_of(x); G.p2_pack([(x, False, dst, am)]); return G.P2Image(dst, am)
M = 16384
res = []
for N, K in ((3072, 512), (3072, 32), (1024, 512), (512, 512)):
    A, W, C = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.empty(M, N, device=dev)
    Ai, Wi = pack(A), pack(W)
    res.append("N=%d K=%d: %.1f us" % (N, K, timeit(lambda: G.gemm_p2(Ai, Wi, C, M, N, K))))
print("stagger", os.environ.get("LVT_P2_STAGGER", "0"), " | ".join(res))
