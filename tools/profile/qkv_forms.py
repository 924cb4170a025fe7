"""The q/k/v projection of a DSFVT layer (16384 x 512 -> 3 x 4 heads x 128) as the launch the layer issues (3 x 4 batches of N = 128
over the packed (3, na, d, da) weights) against the same product as plain launches: python tools/profile/qkv_forms.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L, gemm as G
L.set_math_mode("f16x2")
dev = torch.device("cuda:0")
def timeit(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M, d, na, da = 16384, 512, 4, 128
hd = na * da
xn = torch.randn(M, d, device=dev)
w = torch.randn(3, na, d, da, device=dev) * 0.05
qkv = torch.empty(3, M, hd, device=dev)
def layer_form():
    G.gemm(xn, w, qkv, M, da, d, ta=0, tb=1, lda=d, ldb=da, ldc=hd, batch_outer=3, batch_inner=na, sB=(na * d * da, d * da), sC=(M * hd, da))
t0 = timeit(layer_form)
ref = qkv.clone()
# (b) one plain NN launch, N = 1536, weights (d, 3 hd), output (M, 3 hd)
w2 = w.permute(2, 0, 1, 3).reshape(d, 3 * hd).contiguous()
c2 = torch.empty(M, 3 * hd, device=dev)
t1 = timeit(lambda: G.gemm(xn, w2, c2, M, 3 * hd, d, ta=0, tb=1))
e1 = torch.equal(c2.view(M, 3, hd).permute(1, 0, 2), ref)
# (c) three batches of plain NN launches N = 512 over (3, d, hd) weights into the (3, M, hd) slabs
w3 = w.permute(0, 2, 1, 3).reshape(3, d, hd).contiguous()
c3 = torch.empty(3, M, hd, device=dev)
L.set_amax(w3, L.amax_of(w)); L.set_amax(w2, L.amax_of(w))
t2 = timeit(lambda: G.gemm(xn, w3, c3, M, hd, d, ta=0, tb=1, batch_outer=3, batch_inner=1, sB=(d * hd, 0), sC=(M * hd, 0)))
e2 = torch.equal(c3, ref)
# (d) the same as NT over (3, hd, d)
w4 = w3.transpose(1, 2).contiguous()
L.set_amax(w4, L.amax_of(w))
c4 = torch.empty(3, M, hd, device=dev)
t3 = timeit(lambda: G.gemm(xn, w4, c4, M, hd, d, ta=0, tb=0, batch_outer=3, batch_inner=1, sB=(d * hd, 0), sC=(M * hd, 0)))
e3 = torch.equal(c4, ref)
print("layer form (12 batches of N=128) %.1f us | plain NN N=1536 %.1f us (equal %s) | 3 batches NN N=512 %.1f us (equal %s) | 3 batches NT N=512 %.1f us (equal %s)" % (t0, t1, e1, t2, e2, t3, e3))
