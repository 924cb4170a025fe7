"""Micro-benchmark of the engine at the bench shapes (run on the GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvt_amd.hip import gemm as G, binding as L

dev = "cuda:0"
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

N = 512
res = []
for name, Ci, Co, k, s, p, H in [("K2 4x4s2 128->256", 128, 256, 4, 2, 1, 32), ("K3 3x3 256->256", 256, 256, 3, 1, 1, 16),
                                  ("K4a 3x3 256->128", 256, 128, 3, 1, 1, 16), ("K4b 1x1 128->256", 128, 256, 1, 1, 0, 16),
                                  ("K1 4x4s2 4->128", 4, 128, 4, 2, 1, 64)]:
    g = G.conv_geom(N, 1, H, H, Ci, Co, (1, k, k), (1, s, s), (0, p, p))
    x = torch.randn(N, 1, H, H, Ci, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    wp = G.pack_weight(g, w, Ci, Co)
    y = G.conv_fwd(g, x, wp)
    dy = torch.randn_like(y)
    fl = G.conv_flops(g)
    wq = G.pack_weight_parity(g, w, Ci, Co) if G.fwd_by_parity(g) else None
    t1 = timeit(lambda: G.conv_fwd(g, x, wp, wq=wq))
    wph = G.pack_weight_phases(g, w, Ci, Co) if G.bwd_data_by_phases(g) else None
    wt = G.pack_weight_t(g, w, Ci, Co) if G.bwd_data_as_conv(g) else None
    t2 = timeit(lambda: G.conv_bwd_data(g, dy, wp, wt=wt, wph=wph))
    t3 = timeit(lambda: G.conv_bwd_weight(g, x, dy, Ci, Co))
    print("%-20s fwd %7.1f us %6.1f TF | bwd_data %7.1f us %6.1f TF | bwd_w %7.1f us %6.1f TF" %
          (name, t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9, t3 * 1e3, fl / t3 / 1e9))
# transformer GEMMs at b=64
M = 64 * 256
x = torch.randn(M, 512, device=dev); w = torch.randn(512, 512, device=dev); out = torch.empty(M, 512, device=dev)
t = timeit(lambda: G.gemm(x, w, out, M, 512, 512)); print("NT 16384x512x512   %7.1f us %6.1f TF" % (t * 1e3, 2 * M * 512 * 512 / t / 1e9))
t = timeit(lambda: G.gemm(x, w, out, M, 512, 512, ta=0, tb=1, ldb=512)); print("NN 16384x512x512   %7.1f us %6.1f TF" % (t * 1e3, 2 * M * 512 * 512 / t / 1e9))
dw = torch.empty(512, 512, device=dev)
t = timeit(lambda: G.gemm(x, out, dw, 512, 512, M, ta=1, tb=1, lda=512, ldb=512, splits=16)); print("TN 512x512x16384 s16 %7.1f us %6.1f TF" % (t * 1e3, 2 * M * 512 * 512 / t / 1e9))
q = torch.randn(M, 1024, device=dev); P = torch.empty(64, 8, 256, 256, device=dev)
t = timeit(lambda: G.gemm(q, q, P, 256, 256, 128, lda=1024, ldb=1024, ldc=256, batch_outer=64, batch_inner=8, sA=(256 * 1024, 128), sB=(256 * 1024, 128), sC=(8 * 65536, 65536)))
print("QK^T b64h8          %7.1f us %6.1f TF" % (t * 1e3, 2 * 512 * 256 * 256 * 128 / t / 1e9))
wq = torch.randn(8, 512, 128, device=dev)
t = timeit(lambda: G.gemm(x, wq, q, M, 128, 512, ta=0, tb=1, lda=512, ldb=128, ldc=1024, batch_inner=8, sB=(0, 65536), sC=(0, 128)))
print("QKV proj (8 heads)  %7.1f us %6.1f TF" % (t * 1e3, 2 * M * 1024 * 512 / t / 1e9))
Mb = 8192
xb = torch.randn(Mb, 4096, device=dev); wb = torch.randn(8192, 4096, device=dev); ob = torch.empty(Mb, 8192, device=dev)
t = timeit(lambda: G.gemm(xb, wb, ob, Mb, 8192, 4096), 5); print("NT big 8192x8192x4096 %7.1f us %6.1f TF" % (t * 1e3, 2 * Mb * 8192 * 4096 / t / 1e9))
