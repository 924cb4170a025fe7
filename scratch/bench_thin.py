import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvt_amd.hip import gemm as G
dev = "cuda:0"
x = torch.randn(512, 1, 32, 32, 128, device=dev); w = torch.randn(128, 3, 4, 4, device=dev) * 0.05; b = torch.randn(3, device=dev)
for _ in range(3): G.convT4_fwd(x, w, b, True)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): G.convT4_fwd(x, w, b, True)
e.record(); torch.cuda.synchronize()
print("convT4_fwd %.1f us" % (a.elapsed_time(e) / 20 * 1e3))
