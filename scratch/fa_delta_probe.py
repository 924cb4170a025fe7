"""delta handed from backward A to B (one-pass: dO.O + eps, two-pass: sum p dP) against fp64, and sum_j dK_j (exactly 0 in exact arithmetic)"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvt_amd.hip import binding as L, tx
L.set_math_mode("f16x2")
dev = "cuda:0"
S, da, H, B = 256, 128, 8, 2
hd = H * da
blk = (1, 16, 16); T = math.sqrt(da)
torch.manual_seed(3)
for qs in (0.3, 1.5, 0.05, 0.02):
    qkv = torch.randn(3, B * S, hd, device=dev); qkv[0] *= qs
    qkv[1] += 0.5            # a common component of the keys (what an LN bias / the mean token gives)
    if qs < 0.1:             # near-uniform attention over values that are nearly equal: dP - delta is a small difference
        qkv[2] = 1.0 + (0.01 if qs == 0.05 else 0.001) * torch.randn(B * S, hd, device=dev)
    do = torch.randn(B * S, hd, device=dev) * 1e-4
    dt = torch.zeros(H, 1, device=dev); dh = torch.randn(H, 31, device=dev) * 0.2; dw = torch.randn(H, 31, device=dev) * 0.2
    o, st = tx.attn_fwd_flash(qkv, B, H, S, da, T, dt, dh, dw, blk, False)
    # fp64 reference of delta and dK
    q, k, v = [t.double().view(B, S, H, da).permute(0, 2, 1, 3) for t in qkv]
    from oracle import lvt_oracle as O
    bias = O.rel_position_bias(dt.double().cpu(), dh.double().cpu(), dw.double().cpu(), blk).transpose(0, 1).to(dev)
    sc = q @ k.transpose(2, 3) / T + bias
    P = torch.softmax(sc, -1)
    dO = do.double().view(B, S, H, da).permute(0, 2, 1, 3)
    dP = dO @ v.transpose(2, 3)
    dref = (P * dP).sum(-1)                                   # (B, H, S)
    g = P * (dP - dref.unsqueeze(-1))
    dK = (g.transpose(2, 3) @ q) / T
    for onep in (True, False):
        dqkv = tx.attn_bwd_flash(qkv, do, st, B, H, S, da, T, dt, dh, dw, blk, False, o=o if onep else None)[0]
        ws = L.workspace(1, torch.device(dev), "attn_bwd")
        dl = ws[:B * H * S * 4].view(torch.float32).view(B, H, S).double()
        dk = dqkv[1].double().view(B, S, H, da).permute(0, 2, 1, 3)
        print("q scale %.1f onepass=%s: delta err max %.3e (|delta| max %.3e, sum|p dP| typ %.3e) | dK err %.3e | |sum_j dK_j| max %.3e  fp64 %.1e  (|dK| max %.3e)" % (
            qs, onep, float((dl - dref).abs().max()), float(dref.abs().max()), float((P * dP.abs()).sum(-1).mean()),
            float((dk - dK).abs().max()), float(dk.sum(2).abs().max()), float(dK.sum(2).abs().max()), float(dK.abs().max())))
# homogeneity: dO -> 4 dO must give exactly 4x (per-row power-of-two scales)
for onep in (True, False):
    a = tx.attn_bwd_flash(qkv, do, st, B, H, S, da, T, dt, dh, dw, blk, False, o=o if onep else None)[0].clone()
    b = tx.attn_bwd_flash(qkv, do * 4, st, B, H, S, da, T, dt, dh, dw, blk, False, o=o if onep else None)[0]
    print("homogeneous x4, onepass=%s:" % onep, [bool(torch.equal(a[i] * 4, b[i])) for i in range(3)])
