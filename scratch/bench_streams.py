"""Do independent single-round GEMMs overlap better on two streams?  (run on the GPU box)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvt_amd.hip import gemm as G, binding as L
dev = "cuda:0"
M = 16384
x = torch.randn(M, 512, device=dev); w = torch.randn(512, 512, device=dev)
dy = torch.randn(M, 512, device=dev)
dx = torch.empty(M, 512, device=dev); dw = torch.empty(512, 512, device=dev)
s2 = torch.cuda.Stream()
def seq():
    G.gemm(dy, w, dx, M, 512, 512, ta=0, tb=1, ldb=512)
    G.gemm(dy, x, dw, 512, 512, M, ta=1, tb=1, lda=512, ldb=512, splits=16)
def par():
    s2.wait_stream(torch.cuda.current_stream())
    G.gemm(dy, w, dx, M, 512, 512, ta=0, tb=1, ldb=512)
    with torch.cuda.stream(s2):
        G.gemm(dy, x, dw, 512, 512, M, ta=1, tb=1, lda=512, ldb=512, splits=16)
    torch.cuda.current_stream().wait_stream(s2)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
# note: the "gemm" workspace is keyed by stream, so the two streams do not share split-K scratch
for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
    print("%-12s %7.1f us per (dX + dW) pair" % (name, timeit(fn)))
# a chain of 6 NT GEMMs (single-round each) vs the same with 256-row M tiles emulated by halving M twice
y = torch.empty(M, 512, device=dev)
def chain():
    for _ in range(6): G.gemm(x, w, y, M, 512, 512)
print("6 x NT 16384x512x512: %.1f us each" % (timeit(chain) / 6))
