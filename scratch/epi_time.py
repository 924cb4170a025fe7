"""cost of the residual / mask epilogue forms of the wide GEMM: us per launch"""
import sys, torch
sys.path.insert(0, "/root/repo")
from lvt_amd.hip import binding as L, gemm as G
dev = "cuda:0"
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M = 16384
for N, K in [(512, 512), (3072, 512), (512, 3072)]:
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
    c, r, mk, b = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev), torch.randn(M, N, device=dev), torch.randn(N, device=dev)
    out = [f"N={N} K={K}:"]
    out.append("plain %.1f" % t(lambda: G.gemm(x, w, c, M, N, K)))
    out.append("bias %.1f" % t(lambda: G.gemm(x, w, c, M, N, K, flags=L.EPI_BIAS, bias=b)))
    out.append("res %.1f" % t(lambda: G.gemm(x, w, c, M, N, K, flags=L.EPI_RESIDUAL, res=r)))
    out.append("bias+res %.1f" % t(lambda: G.gemm(x, w, c, M, N, K, flags=L.EPI_BIAS | L.EPI_RESIDUAL, bias=b, res=r)))
    out.append("mask %.1f" % t(lambda: G.gemm(x, w, c, M, N, K, flags=L.EPI_MASK, mask=mk)))
    out.append("bias+relu %.1f" % t(lambda: G.gemm(x, w, c, M, N, K, flags=L.EPI_BIAS | L.EPI_RELU, bias=b)))
    print("  ".join(out), " (extra read: %.0f MB)" % (M * N * 4 / 1e6), flush=True)
