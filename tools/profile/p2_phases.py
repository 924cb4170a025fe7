"""per-workgroup phase timestamps of lvt_gemm_p2_kernel (P2_X_DBG build): where the fixed cost of a tile goes"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lvt_amd.hip import binding as L
from lvt_amd.hip import gemm as G
dev = torch.device("cuda:0")
L.set_math_mode("f16x2")
lib = L.lib()
def pack(x):
    dst = torch.empty_like(x); am = L.amax_of(x); G.p2_pack([(x, False, dst, am)]); return G.P2Image(dst, am)
M = 16384
for N, K in ((512, 32), (512, 512), (3072, 512)):
    A, W, C = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.empty(M, N, device=dev)
    Ai, Wi = pack(A), pack(W)
    nwg = (M // 256) * (N // 128)
    dbg = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
    for _ in range(3): G.gemm_p2(Ai, Wi, C, M, N, K)
    lib.lvt_p2_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    G.gemm_p2(Ai, Wi, C, M, N, K)
    torch.cuda.synchronize()
    lib.lvt_p2_debug_buffer(ctypes.c_void_p(0))
    d = dbg.cpu().double()
    t0 = d[:, 0].min()
    names = ["entry->prologue issued", "->first tile landed (K=32 only)", "->main loop done", "->acc final", "->epilogue stores issued", "->amax"]
    seg = [(0, 1), (1, 2), (1 if K > 32 else 2, 3), (3, 4), (4, 5), (5, 6)]
    print("N=%d K=%d: %d workgroups; cycle counts (median over workgroups); kernel span %.0f cycles" % (N, K, nwg, float(d[:, 6].max() - t0)))
    for nm, (a, b) in zip(names, seg):
        if K > 32 and a == 1 and b == 2: continue
        print("   %-36s %8.0f" % (nm, float((d[:, b] - d[:, a]).median())))
    print("   workgroup total (median) %.0f; start spread of the first 256: %.0f" % (float((d[:, 6] - d[:, 0]).median()), float(d[:256, 0].max() - t0)))
    st = (d[:, 0] - t0).sort().values
    print("   starts (sorted, every 128th):", [int(x) for x in st[::128]])
