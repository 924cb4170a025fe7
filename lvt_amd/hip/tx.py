"""Typed wrappers over the transformer-specific kernels (csrc/transformer.hip + the one-hot GEMM)."""
import ctypes as C

import torch

from . import binding as L
from . import ew


def _iarr(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def attn_softmax_fwd_(scores, temper, dt, dh, dw, block, masked, fill=-1e4):
    L.require(scores, dt, dh, dw)
    B, H, S, _ = scores.shape
    L.check(L.lib().lvt_attn_softmax_fwd(L.ptr(scores), B, H, S, temper, L.ptr(dt), L.ptr(dh), L.ptr(dw),
                                         block[0], block[1], block[2], 1 if masked else 0, fill, L.stream_ptr()),
            "lvt_attn_softmax_fwd")
    L.drop_amax(scores)         # rewritten in place through a raw pointer: the record of max |q k^T| no longer bounds it
    return scores


def attn_fwd_supported(S, da):
    return S == 256 and da == 128


def attn_fwd(q, k, v, B, H, S, da, temper, dt, dh, dw, block, masked, fill=-1e4):
    """Fused scores + bias + mask + softmax + P.V for one 256-token block -> (P (B,H,S,S), o (B*S, H*da))."""
    L.require(q, k, v, dt, dh, dw)
    P = torch.empty(B, H, S, S, dtype=torch.float32, device=q.device)
    o = torch.empty(B * S, H * da, dtype=torch.float32, device=q.device)
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(L.lib().lvt_attn_fwd(L.ptr(q), L.ptr(k), L.ptr(v), B, H, S, da, temper, L.ptr(dt), L.ptr(dh), L.ptr(dw),
                                 block[0], block[1], block[2], 1 if masked else 0, fill, L.ptr(P), L.ptr(o),
                                 L.stream_ptr()), "lvt_attn_fwd")
    if t0 is not None:
        L.TIMER.end("attn_fwd", 4.0 * B * H * S * S * da, t0)
    return P, o


def attn_planes_supported(S, da, block, bh_pairs=8):
    """bh_pairs = batch x heads (the pipelined kernels pair their workgroups per XCD: a multiple of 8)."""
    return bh_pairs % 8 == 0 and bool(L.lib().lvt_attn_planes_supported(S, da, block[0], block[1], block[2])) and L.get_math_mode() != "f32"


def attn_fwd_planes(qkvp, B, H, S, da, temper, dt, dh, dw, block, masked, fill=-1e4):
    """qkvp (3 operands, 3 planes, B*S, H*da) bf16: the exact 3-way split of q, k, v (gemm with EPI_PLANES)
    -> (P (B,H,S,S), o (B*S, H*da)) fp32."""
    L.require(qkvp, dt, dh, dw)
    M, hd = B * S, H * da
    P = torch.empty(B, H, S, S, dtype=torch.float32, device=qkvp.device)
    o = torch.empty(M, hd, dtype=torch.float32, device=qkvp.device)
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(L.lib().lvt_attn_fwd_planes(L.ptr(qkvp), M * hd, 3 * M * hd, B, H, S, da, temper, L.ptr(dt), L.ptr(dh), L.ptr(dw),
                                        block[0], block[1], block[2], 1 if masked else 0, fill, L.ptr(P), L.ptr(o),
                                        L.out_amax(o), L.stream_ptr()), "lvt_attn_fwd_planes")
    if t0 is not None:
        L.TIMER.end("attn_fwd", 4.0 * B * H * S * S * da, t0)
    return P, o


def attn_bwd_planes(qkvp, dop, P, o, B, H, S, da, temper, block, masked):
    """-> (dqkv (3, B*S, H*da) fp32, ddt, ddh, ddw).  dop: (3 planes, B*S, H*da) bf16 split of dO."""
    L.require(qkvp, dop, P, o)
    M, hd = B * S, H * da
    dev = P.device
    dqkv = torch.empty(3, M, hd, dtype=torch.float32, device=dev)
    ddt = torch.empty(H, 2 * block[0] - 1, dtype=torch.float32, device=dev)
    ddh = torch.empty(H, 2 * block[1] - 1, dtype=torch.float32, device=dev)
    ddw = torch.empty(H, 2 * block[2] - 1, dtype=torch.float32, device=dev)
    lib = L.lib()
    nws = lib.lvt_attn_bwd_planes_workspace_bytes(B, H, S, block[0], block[1], block[2])
    ws = L.workspace(nws, dev, "attn_bwd")
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(lib.lvt_attn_bwd_planes(L.ptr(qkvp), M * hd, 3 * M * hd, L.ptr(dop), L.ptr(P), L.ptr(o), B, H, S, da, temper,
                                    block[0], block[1], block[2], 1 if masked else 0, L.ptr(dqkv[0]), L.ptr(dqkv[1]),
                                    L.ptr(dqkv[2]), L.ptr(ddt), L.ptr(ddh), L.ptr(ddw), L.out_amax(dqkv), L.ptr(ws), nws,
                                    L.stream_ptr()),
            "lvt_attn_bwd_planes")
    if t0 is not None:
        L.TIMER.end("attn_bwd", 8.0 * B * H * S * S * da, t0)
    return dqkv, ddt, ddh, ddw


def attn_flash_supported(S, da, block, bh_pairs=8):
    """The flash kernels (csrc/attention_flash.hip) serve 256-token x 128-dim blocks of an instantiated geometry in the f16x2
    arithmetic; bh_pairs = batch x heads must be a multiple of 8 (workgroup pairing per XCD)."""
    return (bh_pairs % 8 == 0 and L.get_math_mode() == "f16x2" and
            bool(L.lib().lvt_attn_flash_supported(S, da, block[0], block[1], block[2])))


def attn_fwd_flash(qkv, B, H, S, da, temper, dt, dh, dw, block, masked, fill=-1e4):
    """qkv (3, B*S, H*da) fp32 (the packed projection output) -> (o (B*S, H*da), stats (2, B*H*S): row max, 1 / row sum).
    No attention matrix is written (vt_attention.py:59-81)."""
    L.require(qkv, dt, dh, dw)
    M, hd = B * S, H * da
    o = torch.empty(M, hd, dtype=torch.float32, device=qkv.device)
    stats = torch.empty(2, B * H * S, dtype=torch.float32, device=qkv.device)
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(L.lib().lvt_attn_fwd_flash(L.ptr(qkv[0]), L.ptr(qkv[1]), L.ptr(qkv[2]), hd, B, H, S, da, temper, L.ptr(dt), L.ptr(dh),
                                       L.ptr(dw), block[0], block[1], block[2], 1 if masked else 0, fill, L.ptr(o), L.ptr(stats),
                                       L.out_amax(o), L.stream_ptr()), "lvt_attn_fwd_flash")
    if t0 is not None:
        L.TIMER.end("attn_fwd", 4.0 * B * H * S * S * da, t0)
    return o, stats


def attn_bwd_flash(qkv, do, stats, B, H, S, da, temper, dt, dh, dw, block, masked, fill=-1e4, o=None):
    """-> (dqkv (3, B*S, H*da) fp32, ddt, ddh, ddw); do (B*S, H*da) fp32.  o: the forward's output (B*S, H*da) -- with it the
    query-stationary launch takes delta_i = dO_i . O_i and makes one pass over the keys (None: two passes)."""
    L.require(qkv, do, stats, dt, dh, dw, o)
    M, hd = B * S, H * da
    dev = qkv.device
    dqkv = torch.empty(3, M, hd, dtype=torch.float32, device=dev)
    ddt = torch.empty(H, 2 * block[0] - 1, dtype=torch.float32, device=dev)
    ddh = torch.empty(H, 2 * block[1] - 1, dtype=torch.float32, device=dev)
    ddw = torch.empty(H, 2 * block[2] - 1, dtype=torch.float32, device=dev)
    lib = L.lib()
    nws = lib.lvt_attn_bwd_flash_workspace_bytes(B, H, S, block[0], block[1], block[2])
    ws = L.workspace(nws, dev, "attn_bwd")
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(lib.lvt_attn_bwd_flash(L.ptr(qkv[0]), L.ptr(qkv[1]), L.ptr(qkv[2]), L.ptr(do), hd, L.ptr(stats), L.ptr(o), B, H, S, da, temper,
                                   L.ptr(dt), L.ptr(dh), L.ptr(dw), block[0], block[1], block[2], 1 if masked else 0, fill,
                                   L.ptr(dqkv[0]), L.ptr(dqkv[1]), L.ptr(dqkv[2]), L.ptr(ddt), L.ptr(ddh), L.ptr(ddw),
                                   L.out_amax(dqkv), L.ptr(ws), nws, L.stream_ptr()), "lvt_attn_bwd_flash")
    if t0 is not None:
        L.TIMER.end("attn_bwd", 8.0 * B * H * S * S * da, t0)      # the reference's four products (the recomputed S / dP are not counted)
    return dqkv, ddt, ddh, ddw


def attn_softmax_bwd_(P, dP, temper, block):
    """dP is overwritten with dS.  Returns (ddt, ddh, ddw)."""
    L.require(P, dP)
    B, H, S, _ = P.shape
    G = torch.empty(H, S, S, dtype=torch.float32, device=P.device)
    ddt = torch.empty(H, 2 * block[0] - 1, dtype=torch.float32, device=P.device)
    ddh = torch.empty(H, 2 * block[1] - 1, dtype=torch.float32, device=P.device)
    ddw = torch.empty(H, 2 * block[2] - 1, dtype=torch.float32, device=P.device)
    L.check(L.lib().lvt_attn_softmax_bwd(L.ptr(P), L.ptr(dP), B, H, S, temper, block[0], block[1], block[2], L.ptr(G),
                                         L.ptr(ddt), L.ptr(ddh), L.ptr(ddw), L.stream_ptr()), "lvt_attn_softmax_bwd")
    L.drop_amax(dP)             # dP now holds dS
    return ddt, ddh, ddw


MAX_SLOTS = 32          # slot tables travel as kernel arguments (BagSlots / KParams.oh_off)


def _embbag_once(idx, bstride, P, rows, slot_off, tab_row, table, D, bias, btable, bindex):
    out = torch.empty(rows, D, dtype=torch.float32, device=idx.device)
    L.check(L.lib().lvt_embbag_fwd(L.ptr(idx), bstride, P, rows, len(slot_off), _iarr(slot_off), _iarr(tab_row),
                                   L.ptr(table), D, L.ptr(bias), L.ptr(btable), L.ptr(bindex), L.ptr(out),
                                   L.stream_ptr()), "lvt_embbag_fwd")
    return out


def embbag_fwd(idx, bstride, P, rows, slot_off, tab_row, table, D, bias=None, btable=None, bindex=None):
    """Bags wider than MAX_SLOTS (the (kt,kh,kw) x nc taps of the DSSVT / DSTSVT context conv) are summed in
    MAX_SLOTS-wide pieces; the piece order is fixed, so the result is reproducible."""
    L.require(idx, table, bias, btable, bindex)
    out = _embbag_once(idx, bstride, P, rows, slot_off[:MAX_SLOTS], tab_row[:MAX_SLOTS], table, D, bias, btable, bindex)
    for s in range(MAX_SLOTS, len(slot_off), MAX_SLOTS):
        part = _embbag_once(idx, bstride, P, rows, slot_off[s:s + MAX_SLOTS], tab_row[s:s + MAX_SLOTS], table, D,
                            None, None, None)
        out = ew.axpy(part, add=out)
    return out


def _onehot_once(idx, V, slot_off, bstride, pstride, P, rows, dout, N, ldb, dense):
    lib = L.lib()
    ns = len(slot_off)
    out = torch.empty(ns * V, N, dtype=torch.float32, device=dout.device)
    nws = lib.lvt_onehot_tn_workspace_bytes(ns, V, N, rows)
    ws = L.workspace(nws, dout.device, "onehot")
    ldb = ldb if ldb is not None else N
    flags = L.math_flag() | (L.ONEHOT_DENSE if dense else 0)
    gather = bool(lib.lvt_onehot_tn_is_gather(ns, V, N, ldb, L.ptr(dout), flags))
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(lib.lvt_onehot_tn_gemm(L.ptr(idx), ns, V, _iarr(slot_off), bstride, pstride, P, rows, L.ptr(dout), ldb, N,
                                   L.ptr(out), flags, L.ptr(L.amax_of(dout)) if (L.f16x2() and not gather) else None,
                                   L.ptr(ws), nws, L.stream_ptr()),
            "lvt_onehot_tn_gemm")
    if t0 is not None:
        if gather:
            L.TIMER.end("embbag_wgrad_gather", 0.0, t0)     # row sums: no matrix-core work to account
        else:
            L.TIMER.end("gemm_onehot_tn", 2.0 * ns * V * N * rows, t0)
    return out


def onehot_tn_gemm(idx, V, slot_off, bstride, pstride, P, rows, dout, N, ldb=None, dense=False):
    """-> (nslots*V, N) gradient of the gathered table (dense=True: always the one-hot GEMM on the matrix cores)."""
    L.require(idx, dout)
    if len(slot_off) <= MAX_SLOTS:
        return _onehot_once(idx, V, slot_off, bstride, pstride, P, rows, dout, N, ldb, dense)
    return torch.cat([_onehot_once(idx, V, slot_off[s:s + MAX_SLOTS], bstride, pstride, P, rows, dout, N, ldb, dense)
                      for s in range(0, len(slot_off), MAX_SLOTS)], 0)



def permute3(x, strides, shape):
    """contiguous `shape` tensor whose element (i0,i1,i2) is x.flatten()[i0*s0+i1*s1+i2*s2]."""
    L.require(x)
    out = torch.empty(*shape, dtype=torch.float32, device=x.device)
    L.check(L.lib().lvt_permute3(L.ptr(x), strides[0], strides[1], strides[2], shape[0], shape[1], shape[2],
                                 L.ptr(out), L.stream_ptr()), "lvt_permute3")
    return out


def xent_fwd(logits, target, tstride_b, tstride_pos, P, ignore, scale, want_rows=False):
    """logits (rows, V); target is a view INTO an int64 tensor (its data_ptr is the (b=0,pos=0) element).
    -> (loss, lse, count) or, with want_rows, (loss, lse, count, row_loss): row_loss[r] = lse - logit[target], 0 where the
    target is the ignore index (the un-normalised terms of the mean)."""
    L.require(logits)
    rows, V = logits.shape
    dev = logits.device
    row_loss = torch.empty(rows, dtype=torch.float32, device=dev)
    lse = torch.empty(rows, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.float32, device=dev)
    lib = L.lib()
    nws = lib.lvt_xent_workspace_bytes()
    ws = L.workspace(nws, dev, "xent")
    L.check(lib.lvt_xent_fwd(L.ptr(logits), C.c_void_p(target.data_ptr()), tstride_b, tstride_pos, P, rows, V, ignore,
                             scale, L.ptr(row_loss), L.ptr(lse), L.ptr(loss), L.ptr(count), L.ptr(ws), nws,
                             L.stream_ptr()), "lvt_xent_fwd")
    if want_rows:
        return loss.view(()), lse, count, row_loss
    return loss.view(()), lse, count


def xent_bwd(logits, target, tstride_b, tstride_pos, P, ignore, lse, count, gout, scale):
    L.require(logits, lse, count, gout)
    rows, V = logits.shape
    dl = torch.empty_like(logits)
    L.check(L.lib().lvt_xent_bwd(L.ptr(logits), C.c_void_p(target.data_ptr()), tstride_b, tstride_pos, P, rows, V,
                                 ignore, L.ptr(lse), L.ptr(count), L.ptr(gout), scale, L.ptr(dl), L.out_amax(dl), L.stream_ptr()),
            "lvt_xent_bwd")
    return dl


def attn_decode(q, Kc, Vc, H, qi, temper, dt, dh, dw, block, ldq=None, pos=None, q_pos=0):
    """q: B rows of H*da floats (row stride ldq, default contiguous); Kc/Vc (B, S, H*da) caches -> o (B, H*da) for
    query position qi over keys 0..qi.  With `pos` (int32 device scalar) the position is read on the device and the
    query rows start q_pos * pos elements after `q`."""
    L.require(Kc, Vc, dt, dh, dw, pos)
    B, S, hd = Kc.shape
    o = torch.empty(B, hd, dtype=torch.float32, device=Kc.device)
    L.check(L.lib().lvt_attn_decode(C.c_void_p(q.data_ptr()), ldq if ldq is not None else hd, L.ptr(Kc), L.ptr(Vc), B, H, S, hd // H, qi, temper, L.ptr(dt), L.ptr(dh),
                                    L.ptr(dw), block[0], block[1], block[2], L.ptr(o), L.ptr(pos), q_pos, L.stream_ptr()),
            "lvt_attn_decode")
    return o


def sample_categorical(logits, temp, u, out, out_stride=1, want_probs=False, pos=None, u_pos=0):
    """Draw one code per row of `logits` (rows, V) with the uniforms `u` (rows,); the int64 codes go to
    out.data_ptr() + row * out_stride (elements).  Returns the probabilities when asked for.
    `pos` (int32 device scalar): the uniforms are read u_pos * pos elements after `u`."""
    L.require(logits, pos)
    rows, V = logits.shape
    probs = torch.empty(rows, V, dtype=torch.float32, device=logits.device) if want_probs else None
    L.check(L.lib().lvt_sample_categorical(L.ptr(logits), rows, V, float(temp), C.c_void_p(u.data_ptr()), C.c_void_p(out.data_ptr()),
                                           out_stride, L.ptr(probs), L.ptr(pos), u_pos, L.stream_ptr()), "lvt_sample_categorical")
    return probs


def decode_gather_codes(codes_ext, nb, pos):
    """codes_ext (rows, S+1) int64, nb (S, taps) int64, pos int32 device scalar -> (rows, taps) codes of the causal
    neighbours of the current position."""
    L.require(codes_ext, nb, pos)
    rows, S1 = codes_ext.shape
    taps = nb.shape[1]
    out = torch.empty(rows, taps, dtype=torch.int64, device=codes_ext.device)
    L.check(L.lib().lvt_decode_gather_codes(L.ptr(codes_ext), L.ptr(nb), L.ptr(pos), rows, S1, taps, L.ptr(out),
                                            L.stream_ptr()), "lvt_decode_gather_codes")
    return out


def decode_commit(codes_ext, pos, drawn=None):
    """codes_ext[:, pos] = drawn (rows,) when given; pos += 1 (both on the device, one launch)."""
    L.require(codes_ext, pos, drawn)
    rows, S1 = codes_ext.shape
    L.check(L.lib().lvt_decode_commit(L.ptr(drawn), rows, S1, L.ptr(codes_ext), L.ptr(pos), L.stream_ptr()),
            "lvt_decode_commit")
