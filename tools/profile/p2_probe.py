"""lvt_gemm_f32 (wide kernel, register staging + in-kernel split) against lvt_gemm_p2_f32 (P2 images staged by LDS-DMA) on the GEMM
shapes of a DSFVT layer at b = 64 (M = 16384).  usage: python tools/profile/p2_probe.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L
from lvt_amd.hip import gemm as G

dev = torch.device("cuda:0")
L.set_math_mode("f16x2")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def timeit(fn, n=reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def r(*s, scale=1.0):
    return torch.randn(*s, device=dev) * scale


def pack(x, transpose=False):
    rows, k = (x.shape[1], x.shape[0]) if transpose else x.shape
    dst = torch.empty(rows, k, device=dev)
    am = L.amax_of(x)
    G.p2_pack([(x, transpose, dst, am)])
    return G.P2Image(dst, am)


M = 16384
print("shape                          engine us (TF)      p2 fp32-A us (TF)    p2 image-A us (TF)")
for name, N, K, tb in [("NT 512x512 (FFN fwd)", 512, 512, 0), ("NT 512x1024 (proj fwd)", 512, 1024, 0), ("NT 512x3072 (dxn)", 512, 3072, 0),
                       ("NN 512x512 (dFFN)", 512, 512, 1), ("NN 1024x512 (dO)", 1024, 512, 1), ("NT 3072x512", 3072, 512, 0)]:
    A, C = r(M, K), torch.empty(M, N, device=dev)
    W = r(N, K, scale=0.05) if tb == 0 else r(K, N, scale=0.05)
    t0 = timeit(lambda: G.gemm(A, W, C, M, N, K, tb=tb, ldb=(K if tb == 0 else N)))
    ref = C.clone()
    Wi = pack(W, transpose=bool(tb))
    t1 = timeit(lambda: G.gemm_p2(A, Wi, C, M, N, K))
    ok1 = torch.equal(C, ref)
    Ai = pack(A)
    t2 = timeit(lambda: G.gemm_p2(Ai, Wi, C, M, N, K))
    ok2 = torch.equal(C, ref)
    fl = 2.0 * M * N * K
    print("%-28s %7.1f (%3.0f)      %7.1f (%3.0f) %s     %7.1f (%3.0f) %s" % (name, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, "==" if ok1 else "!=",
                                                                         t2, fl / t2 / 1e6, "==" if ok2 else "!="))
# the q/k/v forward: 24 batches of (M x 128 x 512) into (3, M, 1024)
d, na, da = 512, 8, 128
hd = na * da
x, w = r(M, d), r(3, na, d, da, scale=0.05)
am = L.amax_of(w)
C = torch.empty(3, M, hd, device=dev)
t0 = timeit(lambda: G.gemm(x, w, C, M, da, d, ta=0, tb=1, lda=d, ldb=da, ldc=hd, batch_outer=3, batch_inner=na, sB=(na * d * da, d * da), sC=(M * hd, da)))
ref = C.clone()
wf = torch.empty(3 * na * da, d, device=dev)
specs = [(w[p_, h_], True, wf[(p_ * na + h_) * da:(p_ * na + h_ + 1) * da], am) for p_ in range(3) for h_ in range(na)]
tp = timeit(lambda: G.p2_pack(specs))
Wi = G.P2Image(wf, am)
kw = dict(lda=d, ldb=d, ldc=hd, batch_outer=3, batch_inner=na, sB=(na * da * d, da * d), sC=(M * hd, da))
t1 = timeit(lambda: G.gemm_p2(x, Wi, C, M, da, d, **kw))
ok1 = torch.equal(C, ref)
xi = pack(x)
t2 = timeit(lambda: G.gemm_p2(xi, Wi, C, M, da, d, **kw))
ok2 = torch.equal(C, ref)
fl = 2.0 * M * 3 * hd * d
print("%-28s %7.1f (%3.0f)      %7.1f (%3.0f) %s     %7.1f (%3.0f) %s   (pack of the 24 weight blocks: %.1f us)" % (
    "QKV fwd 24 x (128 x 512)", t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, "==" if ok1 else "!=", t2, fl / t2 / 1e6, "==" if ok2 else "!=", tp))
# LayerNorm with and without the image
from lvt_amd.hip import ew
xx, lw, lb = r(M, d), r(d) + 1, r(d)
t0 = timeit(lambda: ew.layernorm_fwd(xx, lw, lb))
t1 = timeit(lambda: ew.layernorm_fwd_p2(xx, lw, lb))
print("layernorm_fwd %.1f us, with the P2 image %.1f us" % (t0, t1))
# all weight images of a 16-layer model: 2 images of (3072 + 512 + 512 + 512) x 512 / 1024 matrices per layer
ws = [r(512, 512, scale=0.05) for _ in range(16 * 2)] + [r(512, 1024, scale=0.05) for _ in range(16)]
specs = []
for wt in ws:
    a_ = L.amax_of(wt)
    specs.append((wt, False, torch.empty_like(wt), a_))
    specs.append((wt, True, torch.empty(wt.shape[1], wt.shape[0], device=dev), a_))
tp = timeit(lambda: G.p2_pack(specs), n=10)
print("p2_pack of %d images (%.1f MB of weights, both orientations): %.1f us" % (len(specs), sum(w_.numel() for w_ in ws) * 4 / 1e6, tp))
