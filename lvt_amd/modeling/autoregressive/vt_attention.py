"""Attention building blocks of the latent video transformer
(reference: vidgen/modeling/autoregressive/vt_attention.py:10-202).

Modules keep the reference's parameter / buffer names; BlockLocalAttention executes one whole layer
(LN -> per-head QKV -> QK^T/sqrt(da) + relative-position bias [+ causal fill -1e4] -> softmax -> PV ->
proj + residual -> LN -> Linear -> ReLU -> Linear + residual) as a single autograd node made of
fp32-MFMA GEMM launches and fused HBM-bound kernels on token-major (b*S, d) activations.
"""
import math
import os

import numpy as np
import torch
from torch import nn
from torch.nn import init

from ...hip import binding as L
from ...hip import ew, tx
from ...hip import gemm as G


class PositionalEncoding(nn.Module):
    """3-D sinusoidal position signal (vt_attention.py:10-50).  The signal depends only on the grid
    extents, so it is built once per (T,H,W) on the host with the reference's formula and cached on
    the device as a (T*H*W, d_model) table that one kernel adds to token-major activations."""

    def __init__(self, d_model, num_dims=3, min_timescale=1.0, max_timescale=1.0e4):
        super().__init__()
        assert d_model >= num_dims * 2, "d_model should be >= then 2*num_dims"
        self.d_model, self.num_dims = d_model, num_dims
        self.num_timescales = d_model // (num_dims * 2)
        inc = np.log(max_timescale / min_timescale) / self.num_timescales
        self.register_buffer("inv_timescales",
                             min_timescale * torch.exp(torch.arange(self.num_timescales).float() * -inc))
        self._tables = {}

    def table(self, T, H, W, device):
        key = (T, H, W, str(device))
        if key not in self._tables:
            inv = self.inv_timescales.detach().float().cpu()
            nts = self.num_timescales
            tab = torch.zeros(T, H, W, self.d_model)
            for dim, length in enumerate((T, H, W)):
                scaled = torch.arange(length, dtype=torch.float).view(-1, 1) * inv.view(1, -1)
                sig = torch.cat([torch.sin(scaled), torch.cos(scaled)], 1)       # (length, 2*nts)
                shape = [1, 1, 1, 2 * nts]
                shape[dim] = length
                tab[..., dim * 2 * nts:(dim + 1) * 2 * nts] += sig.view(shape)
            self._tables[key] = tab.view(T * H * W, self.d_model).contiguous().to(device)
        return self._tables[key]

    def add_tokens_(self, x_tok, T, H, W):
        """x_tok (b*T*H*W, d) += signal, in place."""
        return ew.add_periodic_(x_tok, self.table(T, H, W, x_tok.device), T * H * W)

    def forward(self, x):
        """Reference contract: (b, d, T, H, W), in-place add."""
        b, d, T, H, W = x.shape
        tab = self.table(T, H, W, x.device).t().reshape(1, d, T, H, W)
        x += tab
        return x


class MultiHeadAttention(nn.Module):
    """Parameter container (vt_attention.py:84-112): layer_norm, w_q/w_k/w_v (na, d, da), proj."""

    def __init__(self, na, d, da):
        super().__init__()
        self.na, self.da = na, da
        self.layer_norm = nn.LayerNorm(d)
        self.w_q = nn.Parameter(torch.empty(na, d, da))
        self.w_k = nn.Parameter(torch.empty(na, d, da))
        self.w_v = nn.Parameter(torch.empty(na, d, da))
        self.proj = nn.Linear(na * da, d, bias=False)
        self.init_weights()

    def init_weights(self, *args, **kwargs):
        init.xavier_normal_(self.w_q)
        init.xavier_normal_(self.w_k)
        init.xavier_normal_(self.w_v)
        init.xavier_normal_(self.proj.weight)

    def packed_qkv(self):
        """One (3, na, d, da) buffer whose slices ARE w_q / w_k / w_v (the parameters keep their names, shapes and
        state_dict entries): the three projections, their data gradient and their weight gradient then run as one
        engine launch each.  `.to(device)` / a foreign `.data =` breaks the aliasing; it is re-established here."""
        ws = (self.w_q, self.w_k, self.w_v)
        buf = getattr(self, "_wqkv", None)
        n = self.w_q.numel()
        if (buf is None or buf.device != self.w_q.device or
                any(w.data_ptr() != buf.data_ptr() + buf.element_size() * n * i for i, w in enumerate(ws))):
            with torch.no_grad():
                buf = torch.stack([w.detach() for w in ws], 0).contiguous()
                for i, w in enumerate(ws):
                    w.data = buf[i]
            self._wqkv = buf
        return buf


def _splits(tiles, k):
    """split-K factor: one full wave of workgroups (2 resident per CU -> 512 slots), >= 512 rows each.  Measured on
    the 16384-row weight gradients: 512 slots beat 1024 by 12-14 % (tools/ubench/bench_tn.py)."""
    s = max(1, 512 // max(1, tiles))
    return max(1, min(s, k // 512 if k >= 1024 else 1))


P2_IMAGES = None     # None: hip/gemm.py p2_mode() (default "off"; LVT_P2=1 "full", LVT_P2=qkv); True / False: tests force full / off


def prefetch_p2_images(module):
    """Round 6: P2 images (csrc/gemm_p2.hip) of the weights of every attention layer of `module`, in ONE launch per 64 matrices at
    the start of a pass (the weights change once per optimizer step): the packed q/k/v weights as (3 na da, d) rows -- the q/k/v
    projection then stages its weight tiles by LDS-DMA -- and, in the "full" mode, the first FFN weight too (both operands of the
    two products that read a LayerNorm output by LDS-DMA).  Bit-identical, nothing in the step: opt-in (LVT_P2=qkv / LVT_P2=1).  The images ride on the layers (`_p2`) and are
    dropped in the other arithmetic modes."""
    mode = G.p2_mode(P2_IMAGES)
    use, full = mode != "off", mode == "full"
    specs, layers = [], []
    for m in module.modules():
        if not isinstance(m, BlockLocalAttention):
            continue
        m._p2 = None
        mha, f1 = m.mha, m.ffn[1]
        na, d, da = mha.w_q.shape
        if not use or not mha.w_q.is_cuda or d % 32 or da % 32 or f1.weight.shape[1] % 32:
            continue
        wqkv = mha.packed_qkv()
        buf = getattr(m, "_p2_buf", None)
        if buf is None or buf[0].device != wqkv.device or buf[0].shape != (3 * na * da, d) or buf[1].shape != f1.weight.shape:
            buf = m._p2_buf = (torch.empty(3 * na * da, d, dtype=torch.float32, device=wqkv.device),
                               torch.empty_like(f1.weight, dtype=torch.float32))
        aq = L.amax_of(wqkv)
        # (3 na) blocks of (d, da) -> image rows (p, h, j), k = d: one batched entry (building 24 views per layer cost the host
        # 1.7 ms per pass, more than the launches gain)
        specs.append((wqkv.view(3 * na * d, da)[:d], True, buf[0][:da], aq, 3 * na, d * da, da * d))
        i1 = None
        if full:
            a1 = L.amax_of(f1.weight)
            specs.append((f1.weight.detach(), False, buf[1], a1))
            i1 = G.P2Image(buf[1], a1)
        layers.append((m, G.P2Image(buf[0], aq), i1))
    if specs:
        G.p2_pack(specs)
    for m, iq, i1 in layers:
        m._p2 = (iq, i1)


def linear_wgrad(dy, x, n_out, k_in, rows, want_bias=False):
    """dW (n_out, k_in) = dy^T x with deterministic split-K; want_bias additionally returns db = column sums of dy,
    accumulated by the same launch from the dy tiles it streams."""
    dw = torch.empty(n_out, k_in, dtype=torch.float32, device=dy.device)
    tiles = -(-n_out // 128) * -(-k_in // 128)
    splits = _splits(tiles, rows)
    if not want_bias:
        G.gemm(dy, x, dw, n_out, k_in, rows, ta=1, tb=1, lda=dy.shape[-1], ldb=x.shape[-1], splits=splits)
        return dw
    if splits < 2:
        G.gemm(dy, x, dw, n_out, k_in, rows, ta=1, tb=1, lda=dy.shape[-1], ldb=x.shape[-1], splits=splits)
        return dw, G.colsum(dy, rows, n_out)
    db = torch.empty(n_out, dtype=torch.float32, device=dy.device)
    G.gemm(dy, x, dw, n_out, k_in, rows, ta=1, tb=1, lda=dy.shape[-1], ldb=x.shape[-1], splits=splits, a_colsum=db)
    return dw, db


def linear_wgrad_pair(dy0, x0, dy1, x1, n_out, k_in, rows):
    """The weight + bias gradients of TWO linear layers of the same shape as one split-K launch: (dw (2, n_out, k_in),
    db (2, n_out)).  The operands are unrelated allocations, so the batch strides are their address differences (the
    engine's batch strides are plain element offsets).  16 tiles x 2 x 16 k ranges of 1024 rows fill the chip like
    16 x 32 ranges of 512 did, with half the prologues, epilogues and partial sums: the 512 x 512 x 16384 product runs at
    ~130 TFLOP/s alone and the pair at the ~170 of the 512 x 1024 shape (profiles/r03_gemm_shape_and_power_probes.txt)."""
    es = dy0.element_size()
    da, dx = dy1.data_ptr() - dy0.data_ptr(), x1.data_ptr() - x0.data_ptr()
    if (da % (4 * es) or dx % (4 * es) or dy0.shape != dy1.shape or x0.shape != x1.shape or not
            (dy0.is_contiguous() and dy1.is_contiguous() and x0.is_contiguous() and x1.is_contiguous())):
        a, ab = linear_wgrad(dy0, x0, n_out, k_in, rows, want_bias=True)
        b, bb = linear_wgrad(dy1, x1, n_out, k_in, rows, want_bias=True)
        return torch.stack([a, b]), torch.stack([ab, bb])
    dw = torch.empty(2, n_out, k_in, dtype=torch.float32, device=dy0.device)
    db = torch.empty(2, n_out, dtype=torch.float32, device=dy0.device)
    tiles = 2 * -(-n_out // 128) * -(-k_in // 128)
    splits = _splits(tiles, rows)
    if splits < 2:
        a, ab = linear_wgrad(dy0, x0, n_out, k_in, rows, want_bias=True)
        b, bb = linear_wgrad(dy1, x1, n_out, k_in, rows, want_bias=True)
        return torch.stack([a, b]), torch.stack([ab, bb])
    # (each operand of this launch spans two tensors: a_also / b_also bring the second one's max |.| into the f16x2 scale)
    G.gemm(dy0, x0, dw, n_out, k_in, rows, ta=1, tb=1, lda=n_out, ldb=k_in, batch_inner=2, sA=(0, da // es), sB=(0, dx // es),
           sC=(0, n_out * k_in), splits=splits, a_colsum=db, a_also=dy1, b_also=x1)
    return dw, db


CAUSAL_SKIP = not os.environ.get("LVT_NO_CAUSAL_SKIP")      # A/B switch of the causal reductions in the backward products
FUSED_ATTENTION = True      # scores + bias + mask + softmax + P.V in one launch when the block is 256 tokens x 128 dims
# q / k / v / dO as bf16x3 planes into the pipelined attention kernels (csrc/attention_pipe.hip): forward and the whole core
# backward (dQ, dK, dV, bank gradients) as three fused launches on 16-wide tiles (eight waves per workgroup, two per SIMD):
# 147 + 184 + 151 us per layer at the DSFVT shape against 176 + ~410 us for the round-2 path (fused forward, four batched
# GEMMs + softmax-bwd + bank kernel).  Used whenever the block has an instantiation ((1,16,16): DSFVT / KDSFVT; (4,8,8): DSSVT /
# DSTSVT), 256 tokens x 128 head dims, bf16x3 arithmetic and batch x heads % 8 == 0.  LVT_NO_PLANE_ATTENTION=1 (or
# PLANE_ATTENTION = False) keeps the round-2 path; tests force either side.
PLANE_ATTENTION = None
# Round 5: in the f16x2 arithmetic the attention core runs on the flash kernels (csrc/attention_flash.hip): fp32 q / k / v / dO
# straight from the projection GEMMs (no planes), no attention matrix in HBM (two floats per row are saved instead of 256),
# P and dS recomputed in the backward pass.  LVT_NO_FLASH_ATTENTION=1 (or FLASH_ATTENTION = False) keeps the plane kernels.
FLASH_ATTENTION = None
PAIR_FFN_WGRAD = not os.environ.get("LVT_NO_PAIR_WGRAD")     # the two FFN weight gradients of a layer as one 2-batch launch


def _use_flash(S, da, block, pairs):
    if os.environ.get("LVT_NO_FLASH_ATTENTION") or FLASH_ATTENTION is False:
        return False
    return tx.attn_flash_supported(S, da, block, pairs)


def _use_planes(S, da, block, pairs):
    if os.environ.get("LVT_NO_PLANE_ATTENTION") or PLANE_ATTENTION is False:
        return False
    return tx.attn_planes_supported(S, da, block, pairs)


class _BlockLocalAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, block, masked, dt, dh, dw, ln_w, ln_b, w_q, w_k, w_v, proj_w, f0w, f0b, f1w, f1b, f3w, f3b,
                wqkv, p2=None):
        L.require(x)
        M, d = x.shape
        S = block[0] * block[1] * block[2]
        b = M // S
        na, _, da = w_q.shape
        hd = na * da
        temper = math.sqrt(da)
        dev = x.device
        flash = FUSED_ATTENTION and _use_flash(S, da, block, b * na)
        planes = FUSED_ATTENTION and not flash and _use_planes(S, da, block, b * na)
        # p2 = (image of the packed q/k/v weights, image of the first FFN weight) made by prefetch_p2_images: the two products
        # that read a LayerNorm output then take BOTH operands as P2 images (the LayerNorm writes its output a second time, as an
        # image under its a-priori bound) and stage them by LDS-DMA -- same bits as the engine's in-kernel split
        p2 = p2 if (p2 is not None and not planes and L.f16x2()) else None
        p2_full = p2 is not None and p2[1] is not None      # else: the q/k/v weight image only, A stays fp32 (hip/gemm.py p2_mode)
        if p2_full:
            xn, xn_img, mean1, rstd1 = ew.layernorm_fwd_p2(x, ln_w, ln_b)
        else:
            xn, mean1, rstd1 = ew.layernorm_fwd(x, ln_w, ln_b)
        if planes:
            # q, k, v of all heads in ONE launch whose epilogue writes them as their exact 3-way bf16 split (3 operands x
            # 3 planes x (M, hd)): the operand format of the pipelined attention kernels, which then stage by copying
            qkv = torch.empty(3, 3, M, hd, dtype=torch.bfloat16, device=dev)
            G.gemm(xn, wqkv, qkv, M, da, d, ta=0, tb=1, lda=d, ldb=da, ldc=hd, batch_outer=3, batch_inner=na,
                   sB=(na * d * da, d * da), sC=(3 * M * hd, da), flags=L.EPI_PLANES, c_plane=M * hd)
            P, o = tx.attn_fwd_planes(qkv, b, na, S, da, temper, dt, dh, dw, block, masked)
        else:
            # q, k, v of all heads in ONE launch: 3 x na batches of (M x da x d) against the packed weights, C = (3, M, hd)
            qkv = torch.empty(3, M, hd, dtype=torch.float32, device=dev)
            if p2 is not None:
                # (the image rows are (projection, head, j): a projection's heads are ONE (hd, d) matrix -- three batches of N = hd,
                # whose n-tiles share their A panel in one L2, instead of 3 na batches of N = da)
                G.gemm_p2(G.P2Image(xn_img, L.amax_of(xn)) if p2_full else xn, p2[0], qkv, M, hd, d, lda=d, ldb=d, ldc=hd,
                          batch_outer=3, batch_inner=1, sB=(hd * d, 0), sC=(M * hd, 0))
                if p2_full:
                    del xn_img
            else:
                G.gemm(xn, wqkv, qkv, M, da, d, ta=0, tb=1, lda=d, ldb=da, ldc=hd, batch_outer=3, batch_inner=na,
                       sB=(na * d * da, d * da), sC=(M * hd, da))
            q, k, v = qkv[0], qkv[1], qkv[2]
        if planes:
            pass
        elif flash:
            o, P = tx.attn_fwd_flash(qkv, b, na, S, da, temper, dt, dh, dw, block, masked)     # "P": the (2, b*na*S) row statistics
        elif FUSED_ATTENTION and tx.attn_fwd_supported(S, da):
            P, o = tx.attn_fwd(q, k, v, b, na, S, da, temper, dt, dh, dw, block, masked)
        else:
            P = torch.empty(b, na, S, S, dtype=torch.float32, device=dev)
            G.gemm(q, k, P, S, S, da, ta=0, tb=0, lda=hd, ldb=hd, ldc=S, batch_outer=b, batch_inner=na,
                   sA=(S * hd, da), sB=(S * hd, da), sC=(na * S * S, S * S))
            tx.attn_softmax_fwd_(P, temper, dt, dh, dw, block, masked)
            o = torch.empty(M, hd, dtype=torch.float32, device=dev)
            G.gemm(P, v, o, S, da, S, ta=0, tb=1, lda=S, ldb=hd, ldc=hd, batch_outer=b, batch_inner=na,
                   sA=(na * S * S, S * S), sB=(S * hd, da), sC=(S * hd, da))
        y1 = torch.empty(M, d, dtype=torch.float32, device=dev)
        G.gemm(o, proj_w, y1, M, d, hd, flags=L.EPI_RESIDUAL, res=x)
        h1 = torch.empty(M, f1w.shape[0], dtype=torch.float32, device=dev)
        if p2_full:
            fn, fn_img, mean2, rstd2 = ew.layernorm_fwd_p2(y1, f0w, f0b)
            G.gemm_p2(G.P2Image(fn_img, L.amax_of(fn)), p2[1], h1, M, f1w.shape[0], d, flags=L.EPI_BIAS | L.EPI_RELU, bias=f1b)
            del fn_img
        else:
            fn, mean2, rstd2 = ew.layernorm_fwd(y1, f0w, f0b)
            G.gemm(fn, f1w, h1, M, f1w.shape[0], d, flags=L.EPI_BIAS | L.EPI_RELU, bias=f1b)
        if L.RELU_TRACE is not None:
            L.RELU_TRACE.append(h1 > 0)
        y2 = torch.empty(M, d, dtype=torch.float32, device=dev)
        G.gemm(h1, f3w, y2, M, d, f3w.shape[1], flags=L.EPI_BIAS | L.EPI_RESIDUAL, bias=f3b, res=y1)
        ctx.save_for_backward(x, mean1, rstd1, xn, qkv, P, o, y1, mean2, rstd2, fn, h1,
                              ln_w, wqkv, proj_w, f0w, f1w, f3w, dt, dh, dw)
        ctx.block, ctx.dims, ctx.masked, ctx.planes, ctx.flash = block, (M, d, S, b, na, da), bool(masked), bool(planes), bool(flash)
        return y2

    @staticmethod
    def backward(ctx, dy2):
        (x, mean1, rstd1, xn, qkv, P, o, y1, mean2, rstd2, fn, h1,
         ln_w, wqkv, proj_w, f0w, f1w, f3w, dt, dh, dw) = ctx.saved_tensors
        M, d, S, b, na, da = ctx.dims
        hd = na * da
        temper = math.sqrt(da)
        dev = x.device
        dy2 = dy2.contiguous()
        dff = f1w.shape[0]
        # FFN: y2 = h1 W3^T + b3 + y1 ; h1 = relu(fn W1^T + b1)
        dh1 = torch.empty(M, dff, dtype=torch.float32, device=dev)
        G.gemm(dy2, f3w, dh1, M, dff, d, ta=0, tb=1, ldb=dff, flags=L.EPI_MASK, mask=h1)
        dfn = torch.empty(M, d, dtype=torch.float32, device=dev)
        G.gemm(dh1, f1w, dfn, M, d, dff, ta=0, tb=1, ldb=d)
        if d == dff and PAIR_FFN_WGRAD:
            # both FFN weight gradients are (d x d) = dy^T x products over the same rows: one launch
            dwp, dbp = linear_wgrad_pair(dy2, h1, dh1, fn, d, dff, M)
            df3w, df3b, df1w, df1b = dwp[0], dbp[0], dwp[1], dbp[1]
        else:
            df3w, df3b = linear_wgrad(dy2, h1, d, dff, M, want_bias=True)
            df1w, df1b = linear_wgrad(dh1, fn, dff, d, M, want_bias=True)
        dy1, df0w, df0b = ew.layernorm_bwd(dfn, y1, mean2, rstd2, f0w, add=dy2)
        # proj: y1 = o proj^T + x
        if ctx.planes:
            # dO leaves its GEMM as bf16x3 planes; the whole attention core backward is lvt_attn_bwd_planes
            do = torch.empty(3, M, hd, dtype=torch.bfloat16, device=dev)
            G.gemm(dy1, proj_w, do, M, hd, d, ta=0, tb=1, ldb=hd, flags=L.EPI_PLANES, c_plane=M * hd)
            dproj = linear_wgrad(dy1, o, d, hd, M)
            dqkv, ddt, ddh, ddw = tx.attn_bwd_planes(qkv, do, P, o, b, na, S, da, temper, ctx.block, ctx.masked)
            return _BlockLocalAttentionFn._finish_backward(ctx, dqkv, ddt, ddh, ddw, dy1, dproj, df0w, df0b, df1w, df1b,
                                                            df3w, df3b)
        do = torch.empty(M, hd, dtype=torch.float32, device=dev)
        G.gemm(dy1, proj_w, do, M, hd, d, ta=0, tb=1, ldb=hd)
        dproj = linear_wgrad(dy1, o, d, hd, M)
        if ctx.flash:
            # the whole attention core backward on fp32 operands: P and dS are recomputed from q, k, v, dO and the row statistics.
            # `o` selects the one-pass form of the query-stationary launch (delta = dO . O + the row correction): -72 us per
            # unmasked layer at b = 64; the causal layers (12 key chunks per pair instead of 16) run as fast either way (367
            # against 370 us, same box) and take it for its accuracy: the w_q / w_k gradients of the first layers are 2-40x
            # closer to fp64 than with the two-pass form (tests/test_gpu_vt.py, LVT_TEST_VERBOSE=1).
            dqkv, ddt, ddh, ddw = tx.attn_bwd_flash(qkv, do, P, b, na, S, da, temper, dt, dh, dw, ctx.block, ctx.masked, o=o)
            return _BlockLocalAttentionFn._finish_backward(ctx, dqkv, ddt, ddh, ddw, dy1, dproj, df0w, df0b, df1w, df1b,
                                                            df3w, df3b)
        # attention core
        q, k, v = qkv[0], qkv[1], qkv[2]
        bh = dict(batch_outer=b, batch_inner=na)
        dqkv = torch.empty(3, M, hd, dtype=torch.float32, device=dev)
        dq, dk, dv = dqkv[0], dqkv[1], dqkv[2]
        # masked (causal) layers: P and dS vanish above the diagonal (the forward kernel writes exact zeros there), so the
        # products below skip the structurally-zero part of their reductions / tiles
        cz = ctx.masked and CAUSAL_SKIP
        G.gemm(P, do, dv, S, da, S, ta=1, tb=1, lda=S, ldb=hd, ldc=hd, sA=(na * S * S, S * S), sB=(S * hd, da),
               sC=(S * hd, da), flags=L.CAUSAL_KMIN if cz else 0, **bh)              # dV[j] = sum_{i >= j} P[i][j] dO[i]
        dP = torch.empty(b, na, S, S, dtype=torch.float32, device=dev)
        G.gemm(do, v, dP, S, S, da, ta=0, tb=0, lda=hd, ldb=hd, ldc=S, sA=(S * hd, da), sB=(S * hd, da),
               sC=(na * S * S, S * S), flags=L.CAUSAL_TILE if cz else 0, **bh)       # dP[i][j] only matters for j <= i
        ddt, ddh, ddw = tx.attn_softmax_bwd_(P, dP, temper, ctx.block)       # dP now holds dS
        G.gemm(dP, k, dq, S, da, S, ta=0, tb=1, lda=S, ldb=hd, ldc=hd, sA=(na * S * S, S * S), sB=(S * hd, da),
               sC=(S * hd, da), flags=L.CAUSAL_KMAX if cz else 0, **bh)              # dQ[i] = sum_{j <= i} dS[i][j] K[j]
        G.gemm(dP, q, dk, S, da, S, ta=1, tb=1, lda=S, ldb=hd, ldc=hd, sA=(na * S * S, S * S), sB=(S * hd, da),
               sC=(S * hd, da), flags=L.CAUSAL_KMIN if cz else 0, **bh)              # dK[j] = sum_{i >= j} dS[i][j] Q[i]
        del dP
        return _BlockLocalAttentionFn._finish_backward(ctx, dqkv, ddt, ddh, ddw, dy1, dproj, df0w, df0b, df1w, df1b, df3w, df3b)

    @staticmethod
    def _finish_backward(ctx, dqkv, ddt, ddh, ddw, dy1, dproj, df0w, df0b, df1w, df1b, df3w, df3b):
        x, mean1, rstd1, xn = ctx.saved_tensors[:4]
        ln_w, wqkv = ctx.saved_tensors[12], ctx.saved_tensors[13]
        M, d, S, b, na, da = ctx.dims
        hd = na * da
        dev = x.device
        # per-head projections q = xn w_q[h] (k, v alike), all three at once: the data gradient is one GEMM whose
        # reduction runs over (projection, head, da) = 3*hd -- A walks the (3, M, hd) gradient with a 2-level k,
        # B the packed (3, na, d, da) weights -- and the weight gradient one launch of 3 x na batches
        dxn = torch.empty(M, d, dtype=torch.float32, device=dev)
        G.gemm(dqkv, wqkv, dxn, M, d, 3 * hd, ta=0, tb=0, lda=hd, a_kb=hd, a_skb=M * hd, ldb=da, b_kb=da, b_skb=d * da)
        dws = torch.empty(3, na, d, da, dtype=torch.float32, device=dev)
        G.gemm(xn, dqkv, dws, d, da, M, ta=1, tb=1, lda=d, ldb=hd, ldc=da, batch_outer=3, batch_inner=na,
               sB=(M * hd, da), sC=(na * d * da, d * da), splits=_splits(3 * na * -(-d // 128), M))
        dx, dlnw, dlnb = ew.layernorm_bwd(dxn, x, mean1, rstd1, ln_w, add=dy1)
        return (dx, None, None, ddt, ddh, ddw, dlnw, dlnb, dws[0], dws[1], dws[2], dproj, df0w, df0b,
                df1w, df1b, df3w, df3b, None, None)


class BlockLocalAttention(nn.Module):
    def __init__(self, block_size, da, d, n_head, masked=False):
        super().__init__()
        self.block_size = tuple(block_size)
        self.n_head, self.masked = n_head, masked
        self.mha = MultiHeadAttention(n_head, d, da)
        self.ffn = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, d), nn.ReLU(True), nn.Linear(d, d))
        t, h, w = self.block_size
        self.dt_bank = nn.Parameter(torch.zeros(n_head, 2 * t - 1))
        self.dh_bank = nn.Parameter(torch.zeros(n_head, 2 * h - 1))
        self.dw_bank = nn.Parameter(torch.zeros(n_head, 2 * w - 1))
        # index / mask buffers are part of the reference's state_dict (vt_attention.py:146-167); the
        # kernels derive the same indices arithmetically and never read them.
        it = torch.arange(t).view(t, 1, 1).expand(t, h, w).reshape(-1, 1)
        ih = torch.arange(h).view(1, h, 1).expand(t, h, w).reshape(-1, 1)
        iw = torch.arange(w).view(1, 1, w).expand(t, h, w).reshape(-1, 1)
        for name, ix, n in (("dt", it, t), ("dh", ih, h), ("dw", iw, w)):
            self.register_buffer(name, (ix - ix.t() + (n - 1)).reshape(-1))
        if masked:
            s = t * h * w
            self.register_buffer("mask", torch.triu(torch.ones(1, 1, s, s), diagonal=1))
        else:
            self.register_buffer("mask", None)

    def get_B(self):
        """(n_head, 1, S, S) relative-position bias (vt_attention.py:169-174); diagnostic only."""
        t, h, w = self.block_size
        s = t * h * w
        return (self.dt_bank.index_select(1, self.dt) + self.dh_bank.index_select(1, self.dh)
                + self.dw_bank.index_select(1, self.dw)).view(self.n_head, 1, s, s)

    def forward_tokens(self, x_tok, thw=None):
        """(b*T*H*W, d) token-major -> same.  thw = (T, H, W) of the incoming volume; when it differs from the
        block size the tokens are regrouped block by block (vt_attention.py:189-200)."""
        m, f = self.mha, self.ffn
        split = thw is not None and tuple(thw) != self.block_size
        if split:
            perm, inv = _block_permutation(tuple(thw), self.block_size, x_tok.device)
            S = perm.numel()
            x_tok = _RowPermuteFn.apply(x_tok, perm, inv, S)
        y = _BlockLocalAttentionFn.apply(
            x_tok, self.block_size, self.masked, self.dt_bank, self.dh_bank, self.dw_bank,
            m.layer_norm.weight, m.layer_norm.bias, m.w_q, m.w_k, m.w_v, m.proj.weight,
            f[0].weight, f[0].bias, f[1].weight, f[1].bias, f[3].weight, f[3].bias, m.packed_qkv(), getattr(self, "_p2", None))
        if split:
            y = _RowPermuteFn.apply(y, inv, perm, S)
        return y

    def forward(self, x):
        """Reference contract: (B, C, T, H, W) -> same shape."""
        from .. import convstack
        B, C, T, H, W = x.shape
        tok = convstack._TokensIn.apply(x)
        tok = self.forward_tokens(tok, (T, H, W))
        return convstack._TokensOut.apply(tok, B, C, T, H, W)


_perm_cache = {}


def _block_permutation(thw, block, device):
    """Token order of the block-split branch: raster (T,H,W) -> (nt, nh, nw, t, h, w).  Returns (perm, inv) with
    blocked[i] = raster[perm[i]] and raster[j] = blocked[inv[j]] (per sample)."""
    key = (thw, block, str(device))
    hit = _perm_cache.get(key)
    if hit is None:
        (T, H, W), (t, h, w) = thw, block
        if T % t or H % h or W % w:
            raise ValueError("volume %s is not a multiple of the attention block %s" % (thw, block))
        r = torch.arange(T * H * W).view(T // t, t, H // h, h, W // w, w)
        perm = r.permute(0, 2, 4, 1, 3, 5).reshape(-1)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel())
        hit = _perm_cache[key] = (perm.to(device), inv.to(device))
    return hit


class _RowPermuteFn(torch.autograd.Function):
    """Per-sample row gather of a (b*S, d) token matrix (pure data movement; the inverse permutation is its
    own backward, so no scatter / atomics are involved)."""

    @staticmethod
    def forward(ctx, x, perm, inv, S):
        ctx.save_for_backward(perm, inv)
        ctx.S = S
        return ew.row_gather(x.contiguous(), perm, S)

    @staticmethod
    def backward(ctx, g):
        perm, inv = ctx.saved_tensors
        return ew.row_gather(g.contiguous(), inv, ctx.S), None, None, None
