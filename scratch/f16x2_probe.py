"""Engine shapes of the headline step, timed per arithmetic mode (bf16x3 / f16x2 / f32) on random operands.
usage: python scratch/f16x2_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lvt_amd.hip import binding as L
from lvt_amd.hip import gemm as G

dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def r(*s):
    return torch.randn(*s, device=dev)


rows = []
M = 16384
for (N, K, tb, name) in [(512, 512, 0, "ffn NT"), (512, 1024, 0, "proj NT"), (1024, 512, 1, "dO NN"), (512, 3072, 0, "dxn K=3072"),
                         (512, 512, 1, "dfn NN")]:
    A, B, Cc = r(M, K), (r(N, K) if tb == 0 else r(K, N)), torch.empty(M, N, device=dev)
    rows.append((name, 2.0 * M * N * K, lambda A=A, B=B, Cc=Cc, N=N, K=K, tb=tb: G.gemm(A, B, Cc, M, N, K, tb=tb)))
# qkv: 3 x 8 batches of M x 128 x 512
xn, w, qkv = r(M, 512), r(3, 8, 512, 128), torch.empty(3, M, 1024, device=dev)
rows.append(("qkv fwd", 2.0 * M * 3072 * 512, lambda: G.gemm(xn, w, qkv, M, 128, 512, ta=0, tb=1, lda=512, ldb=128, ldc=1024, batch_outer=3,
                                                             batch_inner=8, sB=(8 * 512 * 128, 512 * 128), sC=(M * 1024, 128))))
# weight gradient 512 x 512 over 16384 rows, split-K
dy, x, dw = r(M, 512), r(M, 512), torch.empty(512, 512, device=dev)
rows.append(("wgrad TN 512x512", 2.0 * M * 512 * 512, lambda: G.gemm(dy, x, dw, 512, 512, M, ta=1, tb=1, lda=512, ldb=512, splits=32)))
# convolutions of the VQ-VAE at 512 frames
for (Ci, Co) in [(256, 256), (128, 256)]:
    g = G.conv_geom(512, 1, 16, 16, Ci, Co, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    xx, ww = torch.relu(r(512, 1, 16, 16, Ci)), r(Co, Ci, 1, 3, 3) * 0.05
    gy = r(512, 1, 16, 16, Co)
    rows.append(("conv3x3 fwd %d->%d" % (Ci, Co), G.conv_flops(g), lambda g=g, xx=xx, ww=ww, Ci=Ci, Co=Co: G.conv_fwd(g, xx, G.pack_weight(g, ww, Ci, Co))))
    rows.append(("conv3x3 wgrad %d->%d" % (Ci, Co), G.conv_flops(g), lambda g=g, xx=xx, gy=gy, Ci=Ci, Co=Co: G.conv_bwd_weight(g, xx, gy, Ci, Co)))
g2 = G.conv_geom(512, 1, 32, 32, 128, 256, (1, 4, 4), (1, 2, 2), (0, 1, 1))
x2, w2 = torch.relu(r(512, 1, 32, 32, 128)), r(256, 128, 1, 4, 4) * 0.05
rows.append(("conv4x4s2 fwd 128->256", G.conv_flops(g2), lambda: G.conv_fwd(g2, x2, None, wq=G.pack_weight_parity(g2, w2, 128, 256))))

print("%-26s %s" % ("shape", "   ".join("%-16s" % m for m in ("bf16x3", "f16x2", "f32"))))
for name, fl, fn in rows:
    out = []
    for mode in ("bf16x3", "f16x2", "f32"):
        L.set_math_mode(mode)
        if mode == "f32" and "4x4" in name:
            out.append("-"); continue
        t = timeit(fn)
        out.append("%6.1f us %6.1f TF" % (t * 1e3, fl / t / 1e9))
    print("%-26s %s" % (name, "   ".join(out)))
print("amax fallbacks:", L.AMAX_FALLBACKS[0])
