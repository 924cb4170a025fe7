"""One conv shape, forward only, a few launches (for rocprofv3 --pmc comparisons)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvt_amd.hip import gemm as G
dev = "cuda:0"
which = sys.argv[1]
shapes = {"K2": (128, 256, 4, 2, 1, 32), "K3": (256, 256, 3, 1, 1, 16)}
Ci, Co, k, s, p, H = shapes[which]
N = 512
g = G.conv_geom(N, 1, H, H, Ci, Co, (1, k, k), (1, s, s), (0, p, p))
x = torch.randn(N, 1, H, H, Ci, device=dev)
w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
wp = G.pack_weight(g, w, Ci, Co)
for _ in range(6):
    y = G.conv_fwd(g, x, wp)
torch.cuda.synchronize()
