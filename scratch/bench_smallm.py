import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvt_amd.hip import gemm as G, binding as L, ew
dev = "cuda:0"
def timeit(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, N, K) in ((64, 512, 512), (64, 1024, 512), (64, 512, 1024), (64, 512, 896), (64, 512, 2048), (16, 512, 512)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
    t = timeit(lambda: G.gemm_small(a, w, c, M, N, K))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): G.gemm_small(a, w, c, M, N, K)
    tg = timeit(lambda: g.replay(), 20) / 50
    print("smallm %dx%dx%d eager %.1f us/launch, in-graph %.1f us/launch (weights %.1f MB)" % (M, N, K, t, tg, N * K * 4 / 1e6))
for pad in (0, 16, 32, 80):
    M, N, K = 64, 512, 512
    a = torch.randn(M, K + pad, device=dev); w = torch.randn(N, K + pad, device=dev); c = torch.empty(M, N, device=dev)
    print("pad %d: %.1f us" % (pad, timeit(lambda: G.gemm_small(a, w, c, M, N, K, lda=K + pad, ldb=K + pad))))
x = torch.randn(64, 512, device=dev); wln = torch.ones(512, device=dev); bln = torch.zeros(512, device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50): ew.layernorm_fwd(x, wln, bln, save_stats=False)
print("layernorm 64x512 in-graph %.1f us" % (timeit(lambda: g.replay(), 20) / 50))
g = torch.cuda.CUDAGraph()
y = torch.empty_like(x)
with torch.cuda.graph(g):
    for _ in range(50): torch.add(x, x, out=y)
print("torch add 64x512 in-graph %.1f us" % (timeit(lambda: g.replay(), 20) / 50))
