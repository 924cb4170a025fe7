"""The backward-data product dx = dy W as the NN form (W [K][N], n-contiguous) against the NT form over a transposed copy of W
(W^T [N][K], k-contiguous), DSFVT shapes: python tools/profile/nn_vs_nt.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L, gemm as G
L.set_math_mode("f16x2")
dev = torch.device("cuda:0")
def timeit(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M = 16384
for K, N in ((512, 512), (1536, 512), (2048, 512), (512, 2048)):
    dy = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); Wt = W.t().contiguous()
    c1 = torch.empty(M, N, device=dev); c2 = torch.empty(M, N, device=dev)
    t_nn = timeit(lambda: G.gemm(dy, W, c1, M, N, K, ta=0, tb=1))
    t_nt = timeit(lambda: G.gemm(dy, Wt, c2, M, N, K, ta=0, tb=0))
    print("K %4d N %4d   NN %.1f us   NT %.1f us   equal %s" % (K, N, t_nn, t_nt, torch.equal(c1, c2)))
