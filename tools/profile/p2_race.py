import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L, gemm as G
DEV = torch.device("cuda:0")
L.set_math_mode("f16x2")
def pack(x):
    dst = torch.empty_like(x); am = L.amax_of(x); G.p2_pack([(x, False, dst, am)]); return G.P2Image(dst, am)
def rep(name, fn, n=8):
    outs = []
    for _ in range(n):
        outs.append(fn())
    bad = [int((o != outs[0]).sum()) for o in outs[1:]]
    print("%-44s repeats differing elements vs first: %s" % (name, bad))
    return outs[0]
for M in (2048, 16384):
    d, na, da = 512, 8, 128
    hd = na * da
    x, w = torch.randn(M, d, device=DEV), torch.randn(3 * hd, d, device=DEV) * 0.05
    xi, wi = pack(x), pack(w)
    def eng():
        C = torch.empty(M, 3 * hd, device=DEV); G.gemm(x, w, C, M, 3 * hd, d); return C
    ref = rep("M=%d engine NT N=3072" % M, eng, 3)
    def a():
        C = torch.full((M, 3 * hd), float("nan"), device=DEV); G.gemm_p2(xi, wi, C, M, 3 * hd, d); return C
    o = rep("M=%d p2 image-A N=3072 one launch" % M, a); print("    == engine:", torch.equal(o, ref))
    def b():
        C = torch.full((M, 3 * hd), float("nan"), device=DEV); G.gemm_p2(x, wi, C, M, 3 * hd, d); return C
    o = rep("M=%d p2 fp32-A N=3072 one launch" % M, b); print("    == engine:", torch.equal(o, ref))
    kw = dict(lda=d, ldb=d, ldc=hd, batch_outer=3, batch_inner=na, sB=(na * da * d, da * d), sC=(M * hd, da))
    def c():
        C = torch.full((3, M, hd), float("nan"), device=DEV); G.gemm_p2(xi, wi, C, M, da, d, **kw); return C
    o = rep("M=%d p2 image-A batched 24 x 128" % M, c); print("    == engine:", torch.equal(o.permute(1, 0, 2).reshape(M, 3 * hd), ref))
    def e():
        C = torch.full((3, M, hd), float("nan"), device=DEV); G.gemm_p2(x, wi, C, M, da, d, **kw); return C
    o = rep("M=%d p2 fp32-A batched 24 x 128" % M, e); print("    == engine:", torch.equal(o.permute(1, 0, 2).reshape(M, 3 * hd), ref))
