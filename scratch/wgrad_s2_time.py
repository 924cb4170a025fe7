"""Time the stride-2 frame-resident weight gradient (f16x2) with the library named by LVT_HIP_LIB.  usage: python scratch/wgrad_s2_time.py"""
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
N = 512
g = G.conv_geom(N, 1, 32, 32, 128, 256, (1, 4, 4), (1, 2, 2), (0, 1, 1))
x = torch.relu(torch.randn(N, 1, 32, 32, 128, device=dev)); dy = torch.randn(N, 1, 16, 16, 256, device=dev)
out = []
for kw in ({}, {"want_bias": True}, {"want_bias": True, "bias_of_x": True}):
    t = timeit(lambda: G.conv_bwd_weight(g, x, dy, 128, 256, **kw))
    out.append("%s %.1f us %.0f TF" % (",".join(kw) or "dw", t * 1e3, G.conv_flops(g) / t / 1e9))
g3 = G.conv_geom(N, 1, 16, 16, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1))
x3 = torch.relu(torch.randn(N, 1, 16, 16, 256, device=dev))
t = timeit(lambda: G.conv_bwd_weight(g3, x3, dy, 256, 256))
out.append("3x3 256->256 %.1f us %.0f TF" % (t * 1e3, G.conv_flops(g3) / t / 1e9))
print(os.path.basename(os.environ.get("LVT_HIP_LIB", "default")), " | ".join(out))
