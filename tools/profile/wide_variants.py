"""Time the DSFVT GEMM shapes (f16x2, wide kernel) with the library named by LVT_HIP_LIB.  usage: python tools/profile/wide_variants.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L
from lvt_amd.hip import gemm as G
dev = torch.device("cuda:0")
L.set_math_mode("f16x2")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
r = lambda *s: torch.randn(*s, device=dev)
M = 16384
out = []
for (N, K, tb, ta, name) in [(512, 512, 0, 0, "NT K=512"), (512, 3072, 0, 0, "NT K=3072"), (3072, 512, 0, 0, "NT N=3072 K=512"), (1536, 512, 0, 0, "NT N=1536 K=512"), (1024, 512, 1, 0, "NN N=1024"), (512, 512, 1, 1, "TN wgrad")]:
    if ta == 0:
        A, B, C = r(M, K), (r(N, K) if tb == 0 else r(K, N)), torch.empty(M, N, device=dev)
        t = timeit(lambda: G.gemm(A, B, C, M, N, K, tb=tb))
        fl = 2.0 * M * N * K
    else:
        dy, x, dw = r(M, 512), r(M, 512), torch.empty(512, 512, device=dev)
        t = timeit(lambda: G.gemm(dy, x, dw, 512, 512, M, ta=1, tb=1, lda=512, ldb=512, splits=32))
        fl = 2.0 * M * 512 * 512
    out.append("%s %.1f us %.0f TF" % (name, t * 1e3, fl / t / 1e9))
print(os.path.basename(os.environ.get("LVT_HIP_LIB", "default")), " | ".join(out))
