"""1x1 convolution 256 -> 256 on 512 x 16 x 16 pixels (ResBlock tail): forward forms, us per launch"""
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
N = 512
x = torch.relu(torch.randn(N, 1, 16, 16, 256, device=dev)); w = torch.randn(256, 256, 1, 1, 1, device=dev) * 0.05
b = torch.randn(256, device=dev); res = torch.randn(N, 1, 16, 16, 256, device=dev); mk = torch.randn(N, 1, 16, 16, 256, device=dev)
g = G.conv_geom(N, 1, 16, 16, 256, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0))
wp = G.pack_weight(g, w, 256, 256)
M = N * 256
print("conv_fwd plain %.1f  bias %.1f  bias+res %.1f  res %.1f" % (t(lambda: G.conv_fwd(g, x, wp)), t(lambda: G.conv_fwd(g, x, wp, bias=b)), t(lambda: G.conv_fwd(g, x, wp, bias=b, res=res)), t(lambda: G.conv_fwd(g, x, wp, res=res))))
x2 = x.view(M, 256); w2 = w.view(256, 256).contiguous(); c = torch.empty(M, 256, device=dev); r2 = res.view(M, 256)
print("gemm NT (weight (Co, Ci)) plain %.1f  bias+res %.1f" % (t(lambda: G.gemm(x2, w2, c, M, 256, 256)), t(lambda: G.gemm(x2, w2, c, M, 256, 256, flags=L.EPI_BIAS | L.EPI_RESIDUAL, bias=b, res=r2))))
wt = w2.t().contiguous()
print("gemm NN (weight (Ci, Co)) plain %.1f  bias+res %.1f" % (t(lambda: G.gemm(x2, wt, c, M, 256, 256, tb=1, ldb=256)), t(lambda: G.gemm(x2, wt, c, M, 256, 256, tb=1, ldb=256, flags=L.EPI_BIAS | L.EPI_RESIDUAL, bias=b, res=r2))))
print("bwd_data (mask) %.1f" % t(lambda: G.conv_bwd_data(g, res, wp, mask=mk)))
