"""Time lvt_attn_fwd_flash / lvt_attn_bwd_flash alone at the bench shape (b=64, 8 heads, 256 tokens, d_head 128), next to the
plane kernels of round 3/4 (python scratch/bench_attn_flash.py [fwd|bwd|both] [noplanes])."""
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


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


torch.manual_seed(0)
qkv = torch.randn(3, M, hd, device=dev)
do = torch.randn(M, hd, device=dev)
dt = torch.zeros(na, 1, device=dev); dh = torch.randn(na, 31, device=dev) * 0.1; dw = torch.randn(na, 31, device=dev) * 0.1
which = sys.argv[1] if len(sys.argv) > 1 else "both"
T = math.sqrt(da)
blk = (1, 16, 16)
for masked in (False, True):
    o, stats = tx.attn_fwd_flash(qkv, b, na, S, da, T, dt, dh, dw, blk, masked)
    if which in ("both", "fwd"):
        print("flash ", "masked" if masked else "full  ", "fwd %.1f us" % timeit(lambda: tx.attn_fwd_flash(qkv, b, na, S, da, T, dt, dh, dw, blk, masked)))
    if which in ("both", "bwd"):
        print("flash ", "masked" if masked else "full  ", "bwd %.1f us" % timeit(lambda: tx.attn_bwd_flash(qkv, do, stats, b, na, S, da, T, dt, dh, dw, blk, masked)),
              "bwd one-pass A %.1f us" % timeit(lambda: tx.attn_bwd_flash(qkv, do, stats, b, na, S, da, T, dt, dh, dw, blk, masked, o=o)))
if "noplanes" not in sys.argv:
    qkvp = torch.stack([planes(qkv[i]) for i in range(3)]).contiguous()
    dop = planes(do).contiguous()
    for masked in (False, True):
        P, o2 = tx.attn_fwd_planes(qkvp, b, na, S, da, T, dt, dh, dw, blk, masked)
        if which in ("both", "fwd"):
            print("planes", "masked" if masked else "full  ", "fwd %.1f us" % timeit(lambda: tx.attn_fwd_planes(qkvp, b, na, S, da, T, dt, dh, dw, blk, masked)))
        if which in ("both", "bwd"):
            print("planes", "masked" if masked else "full  ", "bwd %.1f us" % timeit(lambda: tx.attn_bwd_planes(qkvp, dop, P, o2, b, na, S, da, T, blk, masked)))
        print("   max |o_flash - o_planes| / max|o| (masked=%s): %.2e" % (masked, float((tx.attn_fwd_flash(qkv, b, na, S, da, T, dt, dh, dw, blk, masked)[0] - o2).abs().max() / o2.abs().max())))
