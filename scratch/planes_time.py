"""plane-fed B vs in-kernel split (us per launch), and the cost of making the planes of one layer's weights"""
import sys, torch
sys.path.insert(0, "/root/repo")
from lvt_amd.hip import binding as L, gemm as G
dev = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M = 16384
for N, K in [(512, 512), (3072, 512), (512, 3072), (1536, 512)]:
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
    c = torch.empty(M, N, device=dev)
    hi, lo = L.planes_of(w)
    t0 = t(lambda: G.gemm(x, w, c, M, N, K))
    t1 = t(lambda: G.gemm(x, hi, c, M, N, K, b_lo=lo))
    print(f"NT N={N} K={K}: split in kernel {t0:7.1f} us   planes {t1:7.1f} us", flush=True)
ws = [torch.randn(512, 512, device=dev), torch.randn(3072, 512, device=dev), torch.randn(512, 3072, device=dev), torch.randn(6144, 128, device=dev)] * 16
def mk():
    L.bump_epoch(); L.planes_prefetch(ws)
L.amax_prefetch(ws)
print("planes of 16 layers' weights: %.1f us per pass (%.1f M elements)" % (t(mk, 5), sum(w.numel() for w in ws) / 1e6))
