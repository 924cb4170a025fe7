"""gather vs dense one-hot GEMM on embedding weight-gradient shapes: us per launch (hip events, 20 launches)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from lvt_amd.hip import binding as L, tx
dev = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, P, nv = 16, 1024, 512
for ns, N in [(32, 128), (28, 128), (4, 128), (4, 512), (1, 512), (2, 512), (3, 512), (32, 64), (8, 256)]:
    idx = torch.randint(0, nv, (B, ns, P), device=dev)
    dout = torch.randn(B * P, N, device=dev)
    off = [k * P for k in range(ns)]
    tg = t(lambda: tx.onehot_tn_gemm(idx, nv, off, ns * P, 1, P, B * P, dout, N))
    td = t(lambda: tx.onehot_tn_gemm(idx, nv, off, ns * P, 1, P, B * P, dout, N, dense=True))
    print(f"ns={ns:3d} N={N:4d}  gather {tg:7.1f} us   dense {td:7.1f} us   dout re-read {ns*B*P*N*4/1e6:6.1f} MB -> {ns*B*P*N*4/tg/1e6:5.2f} TB/s", flush=True)
