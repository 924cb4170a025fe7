"""dgrad GEMMs dX = dY @ W (W (out, in) row-major -> B n-contiguous, 'nn') against the same product over a transposed copy of W
('nt'): us per launch, f16x2."""
import sys, torch
sys.path.insert(0, "/root/repo")
from lvt_amd.hip import binding as L, gemm as G, tx
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
for nout, nin in [(3072, 512), (512, 3072), (512, 512), (1536, 512)]:      # weight (out, in); dgrad: N = in, K = out
    dy = torch.randn(M, nout, device=dev); w = torch.randn(nout, nin, device=dev) * 0.05
    wt = w.t().contiguous()
    dx = torch.empty(M, nin, device=dev); dx2 = torch.empty(M, nin, device=dev)
    L.amax_of(dy); L.amax_of(w); L.amax_of(wt)
    tnn = t(lambda: G.gemm(dy, w, dx, M, nin, nout, ta=0, tb=1, ldb=nin))
    tnt = t(lambda: G.gemm(dy, wt, dx2, M, nin, nout, ta=0, tb=0, ldb=nout))
    ttr = t(lambda: w.t().contiguous())
    fl = 2.0 * M * nin * nout
    print(f"W {nout}x{nin}: nn {tnn:7.1f} us {fl/tnn/1e6:6.1f} TF   nt(W^T) {tnt:7.1f} us {fl/tnt/1e6:6.1f} TF   transpose {ttr:5.1f} us   equal {torch.equal(dx, dx2)}", flush=True)
