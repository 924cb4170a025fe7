import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
N, Ci, Co, H = 512, 256, 256, 16
g = G.conv_geom(N, 1, H, H, Ci, Co, (1, 3, 3), (1, 1, 1), (0, 1, 1))
for name, fill in (("random", None), ("zeros", 0.0), ("ones", 1.0), ("random", None)):
    x = torch.randn(N, 1, H, H, Ci, device=dev) if fill is None else torch.full((N, 1, H, H, Ci), fill, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05 if fill is None else torch.full((Co, Ci, 3, 3), fill, device=dev)
    wp = G.pack_weight(g, w, Ci, Co)
    t = timeit(lambda: G.conv_fwd(g, x, wp))
    print("K3 fwd %-7s %7.1f us %6.1f TF" % (name, t * 1e3, G.conv_flops(g) / t / 1e9))
M = 16384
for name, fill in (("random", None), ("zeros", 0.0)):
    x = torch.randn(M, 4096, device=dev) if fill is None else torch.zeros(M, 4096, device=dev)
    w = torch.randn(4096, 4096, device=dev) if fill is None else torch.zeros(4096, 4096, device=dev)
    out = torch.empty(M, 4096, device=dev)
    t = timeit(lambda: G.gemm(x, w, out, M, 4096, 4096), 5)
    print("GEMM NT 16384x4096x4096 %-7s %7.1f us %6.1f TF" % (name, t * 1e3, 2 * M * 4096 * 4096 / t / 1e9))
