import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L, gemm as G
DEV = torch.device("cuda:0")
L.set_math_mode("f16x2")
def pack(x):
    dst = torch.empty_like(x); am = L.amax_of(x); G.p2_pack([(x, False, dst, am)]); return G.P2Image(dst, am)
M, d, N = 16384, 512, 3072
x, w = torch.randn(M, d, device=DEV), torch.randn(N, d, device=DEV) * 0.05
xi, wi = pack(x), pack(w)
ref = torch.empty(M, N, device=DEV); G.gemm(x, w, ref, M, N, d)
for rep in range(3):
    C = torch.full((M, N), float("nan"), device=DEV); G.gemm_p2(xi, wi, C, M, N, d)
    torch.cuda.synchronize()
    bad = (C != ref)
    tiles = bad.view(M // 256, 256, N // 128, 128).any(3).any(1)          # (64, 24) wrong tiles
    print("rep", rep, "nan:", int(torch.isnan(C).sum()), "wrong elements:", int(bad.sum()), "wrong tiles:", int(tiles.sum()), "of", tiles.numel())
    rows = bad.any(1).view(M // 256, 256)
    print("   wrong tiles per m-tile (first 16):", tiles.sum(1)[:16].tolist(), " per n-tile:", tiles.sum(0).tolist())
    t = tiles.nonzero()[0].tolist()
    sub = bad[t[0] * 256:(t[0] + 1) * 256, t[1] * 128:(t[1] + 1) * 128]
    print("   first wrong tile", t, "wrong rows:", sub.any(1).nonzero().flatten().tolist()[:40], "wrong cols count:", int(sub.any(0).sum()))
    cs, rs = C[t[0] * 256:(t[0] + 1) * 256, t[1] * 128:(t[1] + 1) * 128], ref[t[0] * 256:(t[0] + 1) * 256, t[1] * 128:(t[1] + 1) * 128]
    r_ = sub.any(1).nonzero().flatten()[0]
    print("   sample row: got", cs[r_, :4].tolist(), "want", rs[r_, :4].tolist())
