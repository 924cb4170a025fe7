"""Does de-phasing help the K = 512 products?  The same work as 16384-row launches on one stream, and as two chains of
8192-row launches on two streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvt_amd.hip import gemm as G
dev = "cuda:0"
M, N, K, REP = 16384, 512, 512, 40
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
C = torch.empty(M, N, device=dev); C2 = torch.empty(M, N, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one_stream():
    x, y = A, C
    for _ in range(REP):
        G.gemm(x, W, y, M, N, K)
        x, y = y, (C2 if y is C else C)
def two_streams():
    h = M // 2
    for s, lo in ((s1, 0), (s2, h)):
        with torch.cuda.stream(s):
            x, y = A[lo:lo + h], C[lo:lo + h]
            for _ in range(REP):
                G.gemm(x, W, y, h, N, K)
                x, y = y, (C2[lo:lo + h] if y.data_ptr() == C[lo:lo + h].data_ptr() else C[lo:lo + h])
def two_streams_interleaved():
    h = M // 2
    xs = [A[:h], A[h:]]; ys = [C[:h], C[h:]]; zs = [C2[:h], C2[h:]]
    for r in range(REP):
        for i, s in enumerate((s1, s2)):
            with torch.cuda.stream(s):
                G.gemm(xs[i], W, ys[i], h, N, K)
                xs[i], ys[i], zs[i] = ys[i], zs[i], ys[i]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    a.record()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for _ in range(n): fn()
    cur.wait_stream(s1); cur.wait_stream(s2)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for name, fn in (("one stream 16384 rows", one_stream), ("two streams 8192 rows (chain after chain)", two_streams),
                 ("two streams 8192 rows (interleaved enqueue)", two_streams_interleaved), ("one stream 16384 rows", one_stream)):
    t = timeit(fn)
    print("%-46s %8.1f us per 16384x512x512 product  %6.1f TF" % (name, t * 1e3 / REP, 2.0 * M * N * K * REP / t / 1e9))
