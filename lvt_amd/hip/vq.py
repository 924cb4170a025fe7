"""Typed wrappers over the vector-quantiser kernels (csrc/vq.hip)."""
import os

import torch

from . import binding as L


VQ_COARSE = 1 << 17          # include/lvt_hip.h: LVT_VQ_COARSE
_COARSE_DEFAULT = bool(os.environ.get("LVT_VQ_COARSE"))


def nearest(z, codebooks, P, coarse=None):
    """z (rows, ldz) channels-last, codebooks (num, K, D) -> idx int64 (rows/P, num, P).
    coarse=True: the coarse-then-exact search (same exact argmin; opt-in, see csrc/vq.hip for when it pays)."""
    L.require(z, codebooks)
    rows, ldz = z.shape
    num, K, D = codebooks.shape
    idx = torch.empty(rows // P, num, P, dtype=torch.int64, device=z.device)
    lib = L.lib()
    nws = lib.lvt_vq_nearest_workspace_bytes(rows, num, K)
    ws = L.workspace(nws, z.device, "vq_nearest")
    L.check(lib.lvt_vq_nearest(L.ptr(z), rows, ldz, num, D, K, L.ptr(codebooks), L.ptr(idx), P, L.math_flag() | (VQ_COARSE if (coarse if coarse is not None else _COARSE_DEFAULT) else 0),
                               L.ptr(ws), nws, L.stream_ptr()), "lvt_vq_nearest")
    return idx


def nearest_single(z, weight, chunk_rows=1 << 16):
    """z (rows, D) channels-last rows, weight (K, D): ONE codebook as wide as the rows (CODEBOOK.NUM == 1) -> idx int64 (rows,).
    Scores x E^T on the GEMM engine (rows are walked in chunks: the score matrix of a chunk is rows x K floats), then
    lvt_vq_argmax_scores; see csrc/vq_single.hip."""
    from . import gemm as G
    L.require(z, weight)
    rows, D = z.shape
    K = weight.shape[0]
    idx = torch.empty(rows, dtype=torch.int64, device=z.device)
    w = weight.detach()
    scores = torch.empty(min(rows, chunk_rows), K, dtype=torch.float32, device=z.device)
    for r0 in range(0, rows, chunk_rows):
        r1 = min(rows, r0 + chunk_rows)
        G.gemm(z[r0:r1], w, scores, r1 - r0, K, D)
        L.check(L.lib().lvt_vq_argmax_scores(L.ptr(scores), r1 - r0, K, K, L.ptr(w), D, L.ptr(idx[r0:r1]), L.stream_ptr()),
                "lvt_vq_argmax_scores")
    return idx


def gather(idx, codebooks):
    """idx (n, num, P) int64 -> (n*P, num*D) channels-last rows of selected code vectors."""
    L.require(idx, codebooks)
    n, num, P = idx.shape
    _, K, D = codebooks.shape
    out = torch.empty(n * P, num * D, dtype=torch.float32, device=idx.device)
    L.check(L.lib().lvt_vq_gather(L.ptr(idx), L.ptr(codebooks), n * P, num, D, K, P, L.ptr(out), num * D,
                                  L.stream_ptr()), "lvt_vq_gather")
    if L.f16x2():
        L.set_amax(out, L.amax_of(codebooks))        # rows of the codebooks: their max |.| bounds the selection
    return out


def ema_accumulate(idx, z, K):
    """-> stats (num, K, D+1): per-code sums and counts of the rows assigned to each code."""
    L.require(idx, z)
    n, num, P = idx.shape
    rows, ldz = z.shape
    D = ldz // num
    stats = torch.empty(num, K, D + 1, dtype=torch.float32, device=z.device)
    lib = L.lib()
    nws = lib.lvt_vq_ema_workspace_bytes(rows, num, D, K)
    ws = L.workspace(nws, z.device, "ema")
    L.check(lib.lvt_vq_ema_accumulate(L.ptr(idx), L.ptr(z), rows, ldz, num, D, K, P, L.ptr(stats), L.ptr(ws), nws,
                                      L.stream_ptr()), "lvt_vq_ema_accumulate")
    return stats


def ema_finalize(stats, running_size, running_sum, weight, decay=0.99, eps=1e-5):
    L.require(stats, running_size, running_sum, weight)
    num, K, D = weight.shape
    L.check(L.lib().lvt_vq_ema_finalize(L.ptr(stats), num, D, K, decay, eps, L.ptr(running_size),
                                        L.ptr(running_sum), L.ptr(weight), L.stream_ptr()), "lvt_vq_ema_finalize")
    L.drop_amax(weight)         # rewritten in place behind torch's back
