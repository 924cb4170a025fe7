"""Time the frame-resident conv kernels (f16x2) with the library named by LVT_HIP_LIB.  usage: python scratch/conv_variants.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvt_amd.hip import binding as L
from lvt_amd.hip import gemm as G
dev = torch.device("cuda:0")
L.set_math_mode("f16x2")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
r = lambda *s: torch.randn(*s, device=dev)
out = []
for (Ci, Co) in [(256, 256), (128, 256)]:
    g = G.conv_geom(512, 1, 16, 16, Ci, Co, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    xx, ww = torch.relu(r(512, 1, 16, 16, Ci)), r(Co, Ci, 1, 3, 3) * 0.05
    wp = G.pack_weight(g, ww, Ci, Co)
    t = timeit(lambda: G.conv_fwd(g, xx, wp))
    out.append("3x3 %d->%d %.1f us %.0f TF" % (Ci, Co, t * 1e3, G.conv_flops(g) / t / 1e9))
g2 = G.conv_geom(512, 1, 32, 32, 128, 256, (1, 4, 4), (1, 2, 2), (0, 1, 1))
x2, w2 = torch.relu(r(512, 1, 32, 32, 128)), r(256, 128, 1, 4, 4) * 0.05
wq = G.pack_weight_parity(g2, w2, 128, 256)
t = timeit(lambda: G.conv_fwd(g2, x2, None, wq=wq))
out.append("4x4s2 128->256 %.1f us %.0f TF" % (t * 1e3, G.conv_flops(g2) / t / 1e9))
print(os.path.basename(os.environ.get("LVT_HIP_LIB", "default")), " | ".join(out))
