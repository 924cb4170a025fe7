"""LayerNorm backward at the DSFVT shape (16384 x 512, with the residual-gradient add): us per launch"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from lvt_amd.hip import binding as L, ew
dev = "cuda:0"
x, dy, add = torch.randn(16384, 512, device=dev), torch.randn(16384, 512, device=dev), torch.randn(16384, 512, device=dev)
w, b = torch.randn(512, device=dev), torch.randn(512, device=dev)
y, mean, rstd = ew.layernorm_fwd(x, w, b)
def t(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
print(os.path.basename(os.environ.get("LVT_HIP_LIB", "default")), "bwd+add %.1f us  bwd %.1f us  fwd %.1f us" % (
    t(lambda: ew.layernorm_bwd(dy, x, mean, rstd, w, add=add)), t(lambda: ew.layernorm_bwd(dy, x, mean, rstd, w)), t(lambda: ew.layernorm_fwd(x, w, b))))
