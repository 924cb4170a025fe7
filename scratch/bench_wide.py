"""conv forward + NT GEMM only (tile-shape experiments); prints time and a checksum."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvt_amd.hip import gemm as G
dev = "cuda:0"
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
torch.manual_seed(0)
N = 512
for name, Ci, Co, k, s, p, H in [("K2 4x4s2 128->256", 128, 256, 4, 2, 1, 32), ("K3 3x3 256->256", 256, 256, 3, 1, 1, 16)]:
    g = G.conv_geom(N, 1, H, H, Ci, Co, (1, k, k), (1, s, s), (0, p, p))
    x = torch.randn(N, 1, H, H, Ci, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    wp = G.pack_weight(g, w, Ci, Co)
    y = G.conv_fwd(g, x, wp)
    fl = G.conv_flops(g)
    t1 = timeit(lambda: G.conv_fwd(g, x, wp))
    print("%-20s fwd %7.1f us %6.1f TF  sum %.6e abs %.6e" % (name, t1 * 1e3, fl / t1 / 1e9, y.double().sum().item(), y.double().abs().sum().item()))
M = 64 * 256
for (n, kk) in [(512, 512), (2048, 512), (512, 2048)]:
    x = torch.randn(M, kk, device=dev); w = torch.randn(n, kk, device=dev); out = torch.empty(M, n, device=dev)
    t = timeit(lambda: G.gemm(x, w, out, M, n, kk))
    print("NT 16384x%dx%d   %7.1f us %6.1f TF  sum %.6e abs %.6e" % (n, kk, t * 1e3, 2 * M * n * kk / t / 1e9, out.double().sum().item(), out.double().abs().sum().item()))
Mb = 8192
xb = torch.randn(Mb, 4096, device=dev); wb = torch.randn(8192, 4096, device=dev); ob = torch.empty(Mb, 8192, device=dev)
t = timeit(lambda: G.gemm(xb, wb, ob, Mb, 8192, 4096), 5); print("NT big 8192x8192x4096 %7.1f us %6.1f TF  abs %.6e" % (t * 1e3, 2 * Mb * 8192 * 4096 / t / 1e9, ob.double().abs().sum().item()))
