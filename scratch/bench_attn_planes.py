"""Time lvt_attn_fwd_planes / lvt_attn_bwd_planes alone at the bench shape (b=64, 8 heads, 256 tokens, d_head 128)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvt_amd.hip import tx
dev = "cuda:0"
b, na, S, da = 64, 8, 256, 128
M, hd = b * S, na * da


def planes(x):
    p1 = x.bfloat16(); r = x - p1.float(); p2 = r.bfloat16(); p3 = (r - p2.float()).bfloat16()
    return torch.stack([p1, p2, p3])


torch.manual_seed(0)
qkv = torch.randn(3, M, hd, device=dev)
qkvp = torch.stack([planes(qkv[i]) for i in range(3)]).contiguous()
dop = planes(torch.randn(M, hd, device=dev)).contiguous()
dt = torch.zeros(na, 1, device=dev); dh = torch.randn(na, 31, device=dev) * 0.1; dw = torch.randn(na, 31, device=dev) * 0.1
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for masked in (False, True):
    P, o = tx.attn_fwd_planes(qkvp, b, na, S, da, math.sqrt(da), dt, dh, dw, (1, 16, 16), masked)
    for name, fn in (("fwd", lambda: tx.attn_fwd_planes(qkvp, b, na, S, da, math.sqrt(da), dt, dh, dw, (1, 16, 16), masked)),
                     ("bwd", lambda: tx.attn_bwd_planes(qkvp, dop, P, o, b, na, S, da, math.sqrt(da), (1, 16, 16), masked))):
        if which not in ("both", name):
            continue
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        print(os.environ.get("LVT_HIP_LIB", "product")[-24:], "masked" if masked else "full  ", name, "%.1f us" % (a.elapsed_time(e) / 10 * 1e3))
