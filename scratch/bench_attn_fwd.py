import sys, os, math
sys.path.insert(0, "/root/repo")
import torch
from lvt_amd.hip import tx
dev = "cuda:0"
b, na, S, da = 64, 8, 256, 128
q = torch.randn(b * S, na * da, device=dev); k = torch.randn_like(q); v = torch.randn_like(q)
dt = torch.zeros(na, 1, device=dev); dh = torch.randn(na, 31, device=dev) * 0.1; dw = torch.randn(na, 31, device=dev) * 0.1
def run(masked):
    return tx.attn_fwd(q, k, v, b, na, S, da, math.sqrt(da), dt, dh, dw, (1, 16, 16), masked)
for masked in (False, True):
    for _ in range(3): run(masked)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): P, o = run(masked)
    e.record(); torch.cuda.synchronize()
    print("masked" if masked else "full  ", "%.1f us" % (a.elapsed_time(e) / 20 * 1e3), float(o.abs().mean()))
