import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import gemm as G
dev = "cuda:0"
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M = 16384
for (n_out, k_in) in ((512, 512), (512, 1024), (1024, 512)):
    dy = torch.randn(M, n_out, device=dev); x = torch.randn(M, k_in, device=dev); dw = torch.empty(n_out, k_in, device=dev)
    for s in (8, 16, 32, 64):
        t = timeit(lambda: G.gemm(dy, x, dw, n_out, k_in, M, ta=1, tb=1, lda=n_out, ldb=k_in, splits=s))
        print("TN %dx%dx%d splits %2d: %6.1f us %6.1f TF" % (n_out, k_in, M, s, t * 1e3, 2 * M * n_out * k_in / t / 1e9))
