"""LayerNorm forward / backward launch times at the DSFVT shape (16384 x 512): python tools/profile/ln_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L, ew
L.set_math_mode("f16x2")
dev = torch.device("cuda:0")
def timeit(fn, n=300):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M, d = 16384, 512
x, dy, add, w, b = (torch.randn(M, d, device=dev) for _ in range(3)) .__iter__().__next__(), torch.randn(M, d, device=dev), torch.randn(M, d, device=dev), torch.randn(d, device=dev), torch.randn(d, device=dev)
x = torch.randn(M, d, device=dev)
y, mean, rstd = ew.layernorm_fwd(x, w, b)
print("fwd %.1f us   bwd(+add) %.1f us   bwd %.1f us" % (timeit(lambda: ew.layernorm_fwd(x, w, b)), timeit(lambda: ew.layernorm_bwd(dy, x, mean, rstd, w, add=add)),
      timeit(lambda: ew.layernorm_bwd(dy, x, mean, rstd, w))))
