"""One-pass / two-pass flash backward: per-sample results must not depend on the batch they are computed in."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L, tx
L.set_math_mode("f16x2")
dev = "cuda:0"
S, da, H = 256, 128, 8
hd = H * da
blk = (1, 16, 16)
T = math.sqrt(da)
torch.manual_seed(1)
for B, scale in ((4, 1.0), (4, 1e-4)):
    qkv = torch.randn(3, B * S, hd, device=dev) * 0.3
    do = torch.randn(B * S, hd, device=dev) * scale
    dt = torch.randn(H, 1, device=dev) * 0.2; dh = torch.randn(H, 31, device=dev) * 0.2; dw = torch.randn(H, 31, device=dev) * 0.2
    for masked in (False, True):
        o, st = tx.attn_fwd_flash(qkv, B, H, S, da, T, dt, dh, dw, blk, masked)
        for onep in (True, False):
            full = tx.attn_bwd_flash(qkv, do, st, B, H, S, da, T, dt, dh, dw, blk, masked, o=o if onep else None)[0].clone()
            worst = 0.0
            for b in range(B):
                sl = slice(b * S, (b + 1) * S)
                q1 = qkv[:, sl].contiguous(); d1 = do[sl].contiguous()
                o1, st1 = tx.attn_fwd_flash(q1, 1, H, S, da, T, dt, dh, dw, blk, masked)
                assert torch.equal(o1, o[sl])
                one = tx.attn_bwd_flash(q1, d1, st1, 1, H, S, da, T, dt, dh, dw, blk, masked, o=o1 if onep else None)[0]
                for i, n in enumerate("qkv"):
                    e = float((one[i] - full[i, sl]).abs().max() / full[i, sl].abs().max())
                    worst = max(worst, e)
                    if e > 0: print("  B=%d masked=%s onepass=%s sample %d d%s differs: rel %.3e" % (B, masked, onep, b, n, e))
            print("B=%d scale=%g masked=%s onepass=%s worst rel diff %.3e" % (B, scale, masked, onep, worst))
            # and twice the same call
            again = tx.attn_bwd_flash(qkv, do, st, B, H, S, da, T, dt, dh, dw, blk, masked, o=o if onep else None)[0]
            print("   run-to-run bitwise:", torch.equal(again, full))
