"""The eight data products of a DSFVT layer (forward and backward-data; M = 16384, d = 512, hd = 1024, dff = 512) on the engine as
the layer issues them, against lvt_gemm_p2_f32 with an fp32 A and the WEIGHT as a P2 image (LDS-DMA for B only):
python tools/profile/p2_bonly.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L, gemm as G
L.set_math_mode("f16x2")
dev = torch.device("cuda:0")
def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
def image(w_nk):
    """P2 image of a (n, k) fp32 matrix."""
    dst = torch.empty_like(w_nk); am = L.amax_of(w_nk); G.p2_pack([(w_nk, False, dst, am)]); return G.P2Image(dst, am)
M = 16384
tot_e = tot_p = 0.0
# (name, N, K, weight stored as [N][K] (NT on the engine) or [K][N] (NN))
for name, N, K, nn in (("qkv fwd", 3072, 512, True), ("proj fwd", 512, 1024, False), ("ffn1 fwd", 512, 512, False), ("ffn2 fwd", 512, 512, False),
                       ("ffn2 bwd", 512, 512, True), ("ffn1 bwd", 512, 512, True), ("proj bwd", 1024, 512, True), ("qkv bwd", 512, 3072, False)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
    Wst = W.t().contiguous() if nn else W
    if nn: L.set_amax(Wst, L.amax_of(W))
    C1, C2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    Wi = image(W)
    te = timeit(lambda: G.gemm(A, Wst, C1, M, N, K, ta=0, tb=1 if nn else 0))
    tp = timeit(lambda: G.gemm_p2(A, Wi, C2, M, N, K))
    tot_e += te; tot_p += tp
    print("%-9s N %4d K %4d %s   engine %6.1f us   p2 (B image) %6.1f us   equal %s" % (name, N, K, "NN" if nn else "NT", te, tp, torch.equal(C1, C2)))
print("sum: engine %.1f us, p2 %.1f us per layer" % (tot_e, tot_p))
