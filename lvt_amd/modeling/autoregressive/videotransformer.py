"""Latent video transformer (reference: vidgen/modeling/autoregressive/videotransformer.py:11-248).

VTEncoder / VTDecoder / ChannelPredictor / VideoTransformer keep the reference's constructor
arguments, parameter names and `forward(context, slice, slice_idx, mode=...)` contract.  Internally
everything runs token-major on the HIP engine:

  * the encoder's one-hot + Conv3d(nc*nv -> de, (kt,1,1)) is an embedding-bag gather over the packed
    weight (the one-hot tensor is never materialised), and so are the decoder's channel embeddings and
    the one-hot inputs of the channel predictor's U_k; their weight gradients are transposed one-hot
    GEMMs on the matrix cores (deterministic, no float atomics);
  * the causal MaskedConv3d runs on the implicit-GEMM conv engine;
  * every 1x1x1 Conv3d / Linear is a fused GEMM (+bias/ReLU/residual epilogue).
"""
import torch
from torch import nn

from ...hip import binding as L
from ...hip import ew, tx
from ...hip import gemm as G
from .. import convstack
from .autoregressive import Autoregressive
from .build import AUTOREGRESSIVE_REGISTRY
from .vt_attention import BlockLocalAttention, PositionalEncoding, linear_wgrad


class MaskedConv3d(nn.Module):
    """Causal 3-D conv parameter container (vt_utils.py:183-200): front pads (kt-1, kh-1, kw//2), taps
    [:, :, -1, -1, kw//2:] re-zeroed in `weight.data` on every forward, like the reference."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True):
        super().__init__()
        for k in kernel_size:
            assert k % 2 == 1
        self.kernel_size = tuple(kernel_size)
        kt, kh, kw = kernel_size
        self.pad = [kw // 2, kw // 2, kh - 1, 0, kt - 1, 0]
        self.causal = kw // 2
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, padding=0, bias=bias)
        self.conv.weight.data = torch.ones(out_channels, in_channels, kt, kh, kw)

    def rezero_(self):
        if self.causal > 0:
            self.conv.weight.data[:, :, -1, -1, self.causal:].zero_()


# ------------------------------------------------------------------------------------------------
# encoder front end: one-hot Conv3d (as a gather) + slice embedding + 1x1x1 projector
# ------------------------------------------------------------------------------------------------
def _flat_tail(w, offset):
    """1-D view of a contiguous weight starting `offset` floats in (a column block of a row-major matrix; the
    GEMM addresses it through its leading dimension)."""
    return w.reshape(-1)[offset:]


class _EncoderFrontFn(torch.autograd.Function):
    """one-hot -> Conv3d(nc*nv -> de, kernel, stride, bias) -> + slice embedding -> [cat class embedding] ->
    1x1x1 projector, token-major (videotransformer.py:35-57).  Slot (tap, channel) of the bag reads one code;
    pads (negative codes) contribute nothing."""

    @staticmethod
    def forward(ctx, context, slice_idx, class_idx, conv_w, conv_b, slice_emb, class_emb, proj_w, nv, stride):
        L.require(context, slice_idx, class_idx)
        b, nc, T, H, W = context.shape
        de, d = conv_w.shape[0], proj_w.shape[0]
        kt, kh, kw = conv_w.shape[2:]
        st, sh, sw = stride
        KK = kt * kh * kw
        if kh == 1 and kw == 1 and T == kt:
            # DSFVT: one output frame, pointwise in space -> slot (tau, c) indexes the context tensor directly
            thw = (1, H, W)
            P = H * W
            idx, bstride = context, nc * kt * P
            off = [(c * kt + tau) * P for tau in range(kt) for c in range(nc)]
        else:
            # general kernel / stride: im2col of the integer codes (index plumbing, no arithmetic):
            # (b, kt, kh, kw, nc, to, ho, wo), slot = (tap, c)
            idx = context.unfold(2, kt, st).unfold(3, kh, sh).unfold(4, kw, sw).permute(0, 5, 6, 7, 1, 2, 3, 4).contiguous()
            thw = tuple(idx.shape[5:])
            P = thw[0] * thw[1] * thw[2]
            bstride = KK * nc * P
            off = [s_ * P for s_ in range(KK * nc)]
        rows = b * P
        # conv weight (de, nc*nv, kt,kh,kw) -> packed (KK, nc*nv, de): row (tap*nc + c)*nv + code
        wt = tx.permute3(conv_w, (1, KK, nc * nv * KK), (KK, nc * nv, de))
        tab = [s_ * nv for s_ in range(KK * nc)]
        e = tx.embbag_fwd(idx, bstride, P, rows, off, tab, wt, de, bias=conv_b, btable=slice_emb, bindex=slice_idx)
        z = torch.empty(rows, d, dtype=torch.float32, device=e.device)
        ce = None
        if class_idx is None:
            G.gemm(e, proj_w, z, rows, d, de)
        else:
            # cat([x, class_emb], channel) @ W^T == x @ W[:, :de]^T + class_emb[cls] @ W[:, de:]^T
            z1 = torch.empty(rows, d, dtype=torch.float32, device=e.device)
            G.gemm(e, proj_w, z1, rows, d, de, ldb=2 * de)
            ce = tx.embbag_fwd(class_idx.repeat_interleave(P).contiguous(), 0, rows, rows, [0], [0], class_emb, de)
            G.gemm(ce, _flat_tail(proj_w, de), z, rows, d, de, ldb=2 * de, flags=L.EPI_RESIDUAL, res=z1)
        ctx.save_for_backward(idx, slice_idx, class_idx, e, ce, proj_w)
        ctx.geo = (b, nc, (kt, kh, kw), P, de, d, nv, off, bstride, slice_emb.shape[0],
                   class_emb.shape[0] if class_emb is not None else 0)
        return z

    @staticmethod
    def backward(ctx, dz):
        idx, slice_idx, class_idx, e, ce, proj_w = ctx.saved_tensors
        b, nc, (kt, kh, kw), P, de, d, nv, off, bstride, n_slices, n_classes = ctx.geo
        KK = kt * kh * kw
        rows = b * P
        dz = dz.contiguous()
        ldp = de if class_idx is None else 2 * de
        de_ = torch.empty(rows, de, dtype=torch.float32, device=dz.device)
        G.gemm(dz, proj_w, de_, rows, de, d, ta=0, tb=1, ldb=ldp)
        dproj = linear_wgrad(dz, e, d, de, rows)
        dclass = None
        if class_idx is not None:
            dce = torch.empty(rows, de, dtype=torch.float32, device=dz.device)
            G.gemm(dz, _flat_tail(proj_w, de), dce, rows, de, d, ta=0, tb=1, ldb=ldp)
            dproj = torch.cat([dproj, linear_wgrad(dz, ce, d, de, rows)], 1)
            dclass = tx.onehot_tn_gemm(class_idx, (n_classes + 3) // 4 * 4, [0], 1, 0, P, rows, dce, de)[:n_classes].contiguous()
        dproj = dproj.view(d, ldp, 1, 1, 1)
        dbias = G.colsum(de_, rows, de)
        # slice embedding: one index per sample (position stride 0)
        dslice = tx.onehot_tn_gemm(slice_idx, (n_slices + 3) // 4 * 4, [0], 1, 0, P, rows, de_, de)[:n_slices]
        dwt = tx.onehot_tn_gemm(idx, nv, off, bstride, 1, P, rows, de_, de)                  # (KK*nc*nv, de)
        dconv = tx.permute3(dwt, (1, de, nc * nv * de), (de, nc * nv, KK)).view(de, nc * nv, kt, kh, kw)
        return None, None, None, dconv, dbias, dslice.contiguous(), dclass, dproj, None, None


class VTEncoder(nn.Module):
    def __init__(self, nc, nv, da, de, d, blocks, n_heads, kernel_size, stride, pad_value=-1, class_num=0):
        super().__init__()
        self.nc, self.nv, self.stride, self.pad_value, self.class_num = nc, nv, tuple(stride), pad_value, class_num
        self.kernel_size = tuple(kernel_size)
        self.conv = nn.Conv3d(nc * nv, de, kernel_size, stride, bias=True)
        self.positional_encoder = PositionalEncoding(de)      # constructed but never applied (reference quirk)
        self.block_local_attention = nn.Sequential(*[BlockLocalAttention(blk, da, d, nh, masked=False)
                                                     for blk, nh in zip(blocks, n_heads)])
        st, sh, sw = stride
        self.slice_embedding = nn.Embedding(st * sh * sw, de)
        if class_num > 0:
            self.class_embedding = nn.Embedding(class_num, de)
            self.linear_projector = nn.Conv3d(2 * de, d, 1, bias=False)
        else:
            self.linear_projector = nn.Conv3d(de, d, 1, bias=False)

    def out_thw(self, context_shape):
        T, H, W = context_shape[2:]
        (kt, kh, kw), (st, sh, sw) = self.kernel_size, self.stride
        return ((T - kt) // st + 1, (H - kh) // sh + 1, (W - kw) // sw + 1)

    def forward_tokens(self, context, slice_idx, class_idx=None):
        """context (b, nc, T', H', W') int64 (already shifted / padded for the conv) -> token-major (b*t*h*w, d)."""
        if self.pad_value >= 0:
            raise L.LvtError("PAD_VALUE must be negative (the gather skips negative codes)")
        use_class = self.class_num > 0 and class_idx is not None
        z = _EncoderFrontFn.apply(context.contiguous(), slice_idx.contiguous(),
                                  class_idx.contiguous() if use_class else None, self.conv.weight, self.conv.bias,
                                  self.slice_embedding.weight, self.class_embedding.weight if use_class else None,
                                  self.linear_projector.weight, self.nv, self.stride)
        thw = self.out_thw(context.shape)
        for layer in self.block_local_attention:
            z = layer.forward_tokens(z, thw)
        return z

    def forward(self, x, slice_idx, class_idx=None):
        b = x.shape[0]
        z = self.forward_tokens(x, slice_idx, class_idx)
        t, h, w = self.out_thw(x.shape)
        return convstack._TokensOut.apply(z, b, z.shape[-1], t, h, w)


# ------------------------------------------------------------------------------------------------
# decoder front end: channel-embedding sum + causal conv + positions + projection of z_l
# ------------------------------------------------------------------------------------------------
class _DecoderFrontFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sl, zl, tables, conv_w, conv_b, proj_w, pos_table, thw):
        L.require(sl, zl)
        b, nc, t, h, w = sl.shape
        nv, de = tables.shape[0] // nc, tables.shape[1]
        d = conv_w.shape[0]
        P = t * h * w
        rows = b * P
        off = [k * P for k in range(nc)]
        emb = tx.embbag_fwd(sl, nc * P, P, rows, off, [k * nv for k in range(nc)], tables, de)
        kt, kh, kw = conv_w.shape[2:]
        # the causal conv pads kt-1 frames in front: with t frames only the last min(kt, t) temporal taps ever see
        # data (DSFVT: t == 1 -> 9 of the 27 taps); the others multiply zero padding and are skipped exactly
        kt_eff = min(kt, t)
        w_eff = conv_w if kt_eff == kt else conv_w[:, :, kt - kt_eff:].contiguous()
        if w_eff is not conv_w and L.f16x2():
            L.set_amax(w_eff, L.amax_of(conv_w))         # (a part of the weight: the whole's max |.| bounds it, no scan)
        g = G.conv_geom(b, t, h, w, de, d, (kt_eff, kh, kw), (1, 1, 1), (kt_eff - 1, kh - 1, kw // 2), out=(t, h, w))
        wp = G.pack_weight(g, w_eff, de, d)
        x = G.conv_fwd(g, emb.view(b, t, h, w, de), wp, bias=conv_b).view(rows, d)
        ew.add_periodic_(x, pos_table, P)
        y = torch.empty(rows, d, dtype=torch.float32, device=x.device)
        G.gemm(zl, proj_w, y, rows, d, d, flags=L.EPI_RESIDUAL, res=x)
        ctx.save_for_backward(sl, zl, emb, wp, proj_w)
        ctx.g, ctx.geo, ctx.kt = g, (b, nc, P, nv, de, d, off), kt
        return y

    @staticmethod
    def backward(ctx, dy):
        sl, zl, emb, wp, proj_w = ctx.saved_tensors
        g = ctx.g
        b, nc, P, nv, de, d, off = ctx.geo
        rows = b * P
        dy = dy.contiguous()
        dzl = torch.empty(rows, d, dtype=torch.float32, device=dy.device)
        G.gemm(dy, proj_w, dzl, rows, d, d, ta=0, tb=1, ldb=d)
        dproj = linear_wgrad(dy, zl, d, d, rows).view(d, d, 1, 1, 1)
        dy5 = dy.view(g.N, g.To, g.Ho, g.Wo, d)
        demb = G.conv_bwd_data(g, dy5, wp).view(rows, de)
        dconv, dbias = G.conv_bwd_weight(g, emb.view(g.N, g.Ti, g.Hi, g.Wi, de), dy5, de, d, want_bias=True)
        if dbias is None:
            dbias = G.colsum(dy, rows, d)
        if g.Kt < ctx.kt:               # taps that only ever saw zero padding: gradient exactly 0
            full = torch.zeros(d, de, ctx.kt, g.Kh, g.Kw, dtype=torch.float32, device=dy.device)
            full[:, :, ctx.kt - g.Kt:] = dconv
            dconv = full
        dtab = tx.onehot_tn_gemm(sl, nv, off, nc * P, 1, P, rows, demb, de)
        return None, dzl, dtab, dconv, dbias, dproj, None, None


class VTDecoder(nn.Module):
    def __init__(self, nc, nv, da, de, d, blocks, n_heads):
        super().__init__()
        self.ch_embedder = nn.ModuleList([nn.Embedding(nv, de) for _ in range(nc)])
        self.de, self.nc, self.nv = de, nc, nv
        self.conv = MaskedConv3d(de, d, (3, 3, 3))
        self.positional_encoder = PositionalEncoding(d)
        self.linear_projector = nn.Conv3d(d, d, 1, bias=False)
        self.block_local_attention = nn.Sequential(*[BlockLocalAttention(blk, da, d, nh, masked=True)
                                                     for blk, nh in zip(blocks, n_heads)])

    def forward_tokens(self, sl, zl_tok):
        b, nc, t, h, w = sl.shape
        self.conv.rezero_()
        tables = torch.cat([e.weight for e in self.ch_embedder], dim=0)          # (nc*nv, de)
        pos = self.positional_encoder.table(t, h, w, zl_tok.device)
        y = _DecoderFrontFn.apply(sl.contiguous(), zl_tok, tables, self.conv.conv.weight, self.conv.conv.bias,
                                  self.linear_projector.weight, pos, (t, h, w))
        for layer in self.block_local_attention:
            y = layer.forward_tokens(y, (t, h, w))
        return y

    def forward(self, slice, zl):
        b, nc, t, h, w = slice.shape
        y = self.forward_tokens(slice, convstack._TokensIn.apply(zl))
        return convstack._TokensOut.apply(y, b, y.shape[-1], t, h, w)


# ------------------------------------------------------------------------------------------------
# channel predictor
# ------------------------------------------------------------------------------------------------
class _ChannelPredictorFn(torch.autograd.Function):
    """logits_k = P_k(relu(U_k([LN(y), onehot(slice[:, :k])]))) for k < nc, token-major (rows, nv)."""

    @staticmethod
    def forward(ctx, yl, sl, ln_w, ln_b, nv, *UP):
        L.require(yl, sl)
        rows, d = yl.shape
        b, nc = sl.shape[:2]
        P = rows // b
        y, mean, rstd = ew.layernorm_fwd(yl, ln_w, ln_b)
        us, outs = [], []
        for k in range(nc):
            uw, ub, pw, pb = UP[4 * k:4 * k + 4]
            fin = uw.shape[1]
            res = None
            if k > 0:
                # transposed one-hot part of U_k: (k*nv, d) so that a code selects a contiguous row
                ut = _permute_cols(uw, d, k * nv)
                res = tx.embbag_fwd(sl, nc * P, P, rows, [c * P for c in range(k)], [c * nv for c in range(k)], ut, d)
            u = torch.empty(rows, d, dtype=torch.float32, device=yl.device)
            G.gemm(y, uw, u, rows, d, d, ldb=fin, flags=L.EPI_BIAS | L.EPI_RELU | (L.EPI_RESIDUAL if k else 0),
                   bias=ub, res=res)
            if L.RELU_TRACE is not None:
                L.RELU_TRACE.append(u > 0)
            o = torch.empty(rows, pw.shape[0], dtype=torch.float32, device=yl.device)
            G.gemm(u, pw, o, rows, pw.shape[0], d, flags=L.EPI_BIAS, bias=pb)
            us.append(u)
            outs.append(o)
        ctx.save_for_backward(yl, sl, mean, rstd, y, ln_w, *us, *UP)
        ctx.geo = (rows, d, b, nc, P, nv)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        rows, d, b, nc, P, nv = ctx.geo
        saved = ctx.saved_tensors
        yl, sl, mean, rstd, y, ln_w = saved[:6]
        us = saved[6:6 + nc]
        UP = saved[6 + nc:]
        dev = yl.device
        dy = torch.empty(rows, d, dtype=torch.float32, device=dev)
        grads = []
        for k in range(nc):
            uw, ub, pw, pb = UP[4 * k:4 * k + 4]
            do = douts[k].contiguous()
            nvk = pw.shape[0]
            fin = uw.shape[1]
            du = torch.empty(rows, d, dtype=torch.float32, device=dev)
            G.gemm(do, pw, du, rows, d, nvk, ta=0, tb=1, ldb=d, flags=L.EPI_MASK, mask=us[k])
            dpw, dpb = linear_wgrad(do, us[k], nvk, d, rows, want_bias=True)
            G.gemm(du, uw, dy, rows, d, d, ta=0, tb=1, ldb=fin, flags=L.EPI_ACCUM if k else 0)
            duw = torch.empty(d, fin, dtype=torch.float32, device=dev)
            duw_d, dub = linear_wgrad(du, y, d, d, rows, want_bias=True)
            duw[:, :d] = duw_d
            if k > 0:
                dut = tx.onehot_tn_gemm(sl, nv, [c * P for c in range(k)], nc * P, 1, P, rows, du, d)   # (k*nv, d)
                duw[:, d:] = dut.t()
            grads += [duw, dub, dpw, dpb]
        dyl, dlnw, dlnb = ew.layernorm_bwd(dy, yl, mean, rstd, ln_w)
        return (dyl, None, dlnw, dlnb, None) + tuple(grads)


def _permute_cols(uw, d, ncols):
    """U.weight[:, d:d+ncols] (out, ncols) -> contiguous (ncols, out)."""
    import ctypes as C
    out = torch.empty(ncols, uw.shape[0], dtype=torch.float32, device=uw.device)
    fin = uw.shape[1]
    L.check(L.lib().lvt_permute3(C.c_void_p(uw.data_ptr() + 4 * d), 1, fin, 0, ncols, uw.shape[0], 1, L.ptr(out),
                                 L.stream_ptr()), "lvt_permute3")
    return out


class _TiedOutputFn(torch.autograd.Function):
    """SHARE_EMBEDDINGS (videotransformer.py:152-154, 174-176): logits = x E^T with E a channel embedding table of the decoder
    (`F.linear(out, weight=ch_embedder[k].weight, bias=None)`), token-major (rows, de) -> (rows, nv)."""

    @staticmethod
    def forward(ctx, x, table):
        L.require(x, table)
        ctx.save_for_backward(x, table)
        out = torch.empty(x.shape[0], table.shape[0], dtype=torch.float32, device=x.device)
        G.gemm(x, table, out, x.shape[0], table.shape[0], x.shape[1])
        return out

    @staticmethod
    def backward(ctx, g):
        x, table = ctx.saved_tensors
        g = g.contiguous()
        rows, nv, de = x.shape[0], table.shape[0], table.shape[1]
        dx = torch.empty(rows, de, dtype=torch.float32, device=x.device)
        G.gemm(g, table, dx, rows, de, nv, ta=0, tb=1, ldb=de)
        return dx, linear_wgrad(g, x, nv, de, rows)


class ChannelPredictor(nn.Module):
    def __init__(self, d, nc, nv, de, share_p=True, share_embeddings=False):
        super().__init__()
        self.nc, self.nv, self.share_p, self.share_embeddings = nc, nv, share_p, share_embeddings
        self._tied = None                  # SHARE_EMBEDDINGS: the decoder's channel embedding tables (set by tie_embeddings)
        self.layer_norm = nn.LayerNorm(d)
        self.U = nn.ModuleList([nn.Linear(d + k * nv, d, bias=True) for k in range(nc)])
        self.relu = nn.ReLU(inplace=True)
        if share_p:         # the reference's config default (defaults.py:50; videotransformer.py:121-123): ONE output layer for all channels
            assert not share_embeddings, "does not make sense"                     # (videotransformer.py:122)
            self.P = nn.Linear(d, nv, bias=True)
        elif share_embeddings:  # ONE layer d -> de; the decoder's channel embedding table E_k is the output matrix (:124-125,152-154)
            self.P = nn.Linear(d, de, bias=True)
        else:
            self.P = nn.ModuleList([nn.Linear(d, nv, bias=True) for _ in range(nc)])

    def _P(self, k):
        return self.P if (self.share_p or self.share_embeddings) else self.P[k]

    def tie_embeddings(self, ch_embedder):
        """SHARE_EMBEDDINGS: remember the decoder's embedding modules WITHOUT registering them here (their parameters stay
        the decoder's: state_dict keys as the reference's, whose forward receives them as an argument)."""
        object.__setattr__(self, "_tied", ch_embedder)

    def _tables(self, ch_embedder=None):
        if not self.share_embeddings:
            return None
        emb = ch_embedder if ch_embedder is not None else self._tied
        if emb is None:
            raise L.LvtError("ChannelPredictor(share_embeddings=True) needs the decoder's ch_embedder (videotransformer.py:236)")
        return [e.weight for e in emb]

    def _flat_params(self):
        """(U_k.weight, U_k.bias, P_k.weight, P_k.bias) per channel; with a shared P the same two tensors appear nc times and
        autograd adds up the nc gradients _ChannelPredictorFn returns for them."""
        flat = []
        for k in range(self.nc):
            flat += [self.U[k].weight, self.U[k].bias, self._P(k).weight, self._P(k).bias]
        return flat

    def logits_tokens(self, sl, yl_tok, ch_embedder=None):
        """-> tuple of nc (rows, nv) token-major logits."""
        outs = _ChannelPredictorFn.apply(yl_tok, sl.contiguous(), self.layer_norm.weight, self.layer_norm.bias,
                                         self.nv, *self._flat_params())
        tables = self._tables(ch_embedder)
        if tables is not None:
            outs = tuple(_TiedOutputFn.apply(o, tables[k]) for k, o in enumerate(outs))
        return outs

    def sample_pixel_tokens(self, yl_tok, b, P, pos, temp=1.0, forced_codes=None, return_probs=False):
        """One pixel of every sample: sequentially draw the nc channels (videotransformer.py:161-185).
        yl_tok (b*P, d); pos = flat position inside the slice.  Returns codes (b, nc) int64.
        `forced_codes` (b, nc) replaces the draws (teacher forcing, used to check the per-channel
        probabilities against the reference since torch.multinomial streams are device specific)."""
        d = yl_tok.shape[-1]
        rows = yl_tok.view(b, P, d)[:, pos].contiguous()                      # (b, d)
        return self.sample_from_rows(rows, temp, forced_codes, return_probs)

    def prepare_decode(self):
        """Transposed copies of the one-hot columns of U_k ((k*nv, d) gather tables), refreshed IN PLACE so that
        captured decode graphs keep reading the same buffers; call once per slice before sample_from_rows."""
        d = self.layer_norm.weight.shape[0]
        if getattr(self, "_ut", None) is None:
            self._ut = [None] + [torch.empty(k * self.nv, d, dtype=torch.float32, device=self.U[k].weight.device)
                                 for k in range(1, self.nc)]
        for k in range(1, self.nc):
            self._ut[k].copy_(self.U[k].weight.detach()[:, d:d + k * self.nv].t())

    def sample_from_rows(self, rows, temp=1.0, forced_codes=None, return_probs=False, uniforms=None, pos=None, split_ws=None):
        """rows (b, d): decoder hidden state of ONE position per sample -> codes (b, nc).
        `uniforms` (P, nc, b) with the int32 device cursor `pos`: the draws of position pos[0] come from uniforms[pos[0]]
        (a table filled once per slice: the decode graphs contain no random-number generator)."""
        b, d = rows.shape
        cached = getattr(self, "_ut", None)
        tables = self._tables()
        y, _, _ = ew.layernorm_fwd(rows, self.layer_norm.weight, self.layer_norm.bias, save_stats=False)
        codes = torch.zeros(b, self.nc, 1, dtype=torch.int64, device=rows.device)
        probs = []
        # one uniform per (sample, channel); a draw is then a pure function of (logits, u) -- lvt_sample_categorical
        u = None
        if forced_codes is None and uniforms is None:
            u = torch.rand(self.nc, b, device=rows.device)
        for k in range(self.nc):
            uw, pw = self.U[k].weight, self._P(k).weight
            res = None
            if k > 0:
                ut = cached[k] if cached is not None else _permute_cols(uw, d, k * self.nv)
                res = tx.embbag_fwd(codes, self.nc, 1, b, list(range(k)), [c * self.nv for c in range(k)], ut, d)
            u_ = torch.empty(b, d, dtype=torch.float32, device=y.device)
            # `split_ws`: the caller's split-K scratch (d >= 1024 takes the split-K form, whose default workspace must not be
            # recorded into a hipGraph: incremental.GraphedSliceSampler passes the decoder's own)
            G.gemm_small(y, uw, u_, b, d, d, ldb=uw.shape[1], flags=L.EPI_BIAS | L.EPI_RELU | (L.EPI_RESIDUAL if k else 0),
                         bias=self.U[k].bias, res=res, split_ws=split_ws)
            o = torch.empty(b, pw.shape[0], dtype=torch.float32, device=y.device)
            G.gemm_small(u_, pw, o, b, pw.shape[0], d, flags=L.EPI_BIAS, bias=self._P(k).bias, split_ws=split_ws)
            if tables is not None:      # SHARE_EMBEDDINGS: (b, de) -> (b, nv) through the tied table (videotransformer.py:174-176)
                o_de, o = o, torch.empty(b, self.nv, dtype=torch.float32, device=y.device)
                G.gemm_small(o_de, tables[k].detach(), o, b, self.nv, pw.shape[0], split_ws=split_ws)
            if forced_codes is None:
                # writes codes[:, k, 0] (element stride nc between samples)
                if uniforms is not None:
                    pr = tx.sample_categorical(o, temp, uniforms.view(-1)[k * b:], codes.view(-1)[k:], self.nc,
                                               want_probs=return_probs, pos=pos, u_pos=self.nc * b)
                else:
                    pr = tx.sample_categorical(o, temp, u[k], codes.view(-1)[k:], self.nc, want_probs=return_probs)
                if return_probs:
                    probs.append(pr)
            else:
                probs.append(torch.softmax(o / temp, 1))
                codes[:, k, 0] = forced_codes[:, k].to(codes.device)
        if return_probs:
            return codes[:, :, 0], torch.stack(probs, 1)
        return codes[:, :, 0]

    def forward(self, slice, yl, mode="logits", pixel=None, temp=1.0, ch_embedder=None, target=None):
        b, d, t, h, w = yl.shape
        tok = convstack._TokensIn.apply(yl)
        if ch_embedder is not None and self.share_embeddings:
            self.tie_embeddings(ch_embedder)
        if mode == "logits":
            outs = self.logits_tokens(slice, tok, ch_embedder)
            return [convstack._TokensOut.apply(o, b, self.nv, t, h, w) for o in outs]
        if mode == "sample_pixel":
            ti, hi, wi = pixel
            return self.sample_pixel_tokens(tok, b, t * h * w, (ti * h + hi) * w + wi, temp)
        raise ValueError


@AUTOREGRESSIVE_REGISTRY.register()
class VideoTransformer(Autoregressive):
    @classmethod
    def from_config(cls, cfg, **kwargs):
        v = cfg.MODEL.AUTOREGRESSIVE.VT
        return cls(nc=v.NC, nv=v.NV, kernel_size=v.KERNEL, stride=v.STRIDE, d=v.D, da=v.DA, de=v.DE,
                   blocks_e=v.BLOCKS_E, n_head_e=v.N_HEAD_E, blocks_d=v.BLOCKS_D, n_head_d=v.N_HEAD_D,
                   pad_value=v.PAD_VALUE, share_p=v.SHARE_P, share_embeddings=v.SHARE_EMBEDDINGS,
                   class_num=v.CLASS_NUM)

    def __init__(self, nc, nv, da, de, d, blocks_e, n_head_e, kernel_size, stride, blocks_d, n_head_d, pad_value,
                 share_p, share_embeddings, class_num):
        super().__init__()
        self.nv, self.nc = nv, nc
        self.encoder = VTEncoder(nc, nv, da, de, d, blocks_e, n_head_e, kernel_size, stride, pad_value, class_num)
        self.decoder = VTDecoder(nc, nv, da, de, d, blocks_d, n_head_d)
        self.ch_predictor = ChannelPredictor(d, nc, nv, de, share_p=share_p, share_embeddings=share_embeddings)
        if share_embeddings:
            self.ch_predictor.tie_embeddings(self.decoder.ch_embedder)       # (videotransformer.py:236, 245)

    # token-major fast path used by VideoTransformerModel -----------------------------------------------
    def logits_tokens(self, context, slice, slice_idx, class_idx=None):
        zl = self.encoder.forward_tokens(context, slice_idx, class_idx)
        yl = self.decoder.forward_tokens(slice, zl)
        return self.ch_predictor.logits_tokens(slice, yl)

    def forward(self, context, slice, slice_idx, mode="logits", pixel=None, zl=None, temp=1.0, drop_mask=None,
                class_idx=None):
        """Reference contract: logits as a list of nc (b, nv, t, h, w) tensors; `sample_pixel` returns
        ((b, nc) codes, zl) with zl in the reference's (b, d, t, h, w) layout."""
        b, nc, t, h, w = slice.shape
        if mode == "logits":
            outs = self.logits_tokens(context, slice, slice_idx, class_idx)
            return [convstack._TokensOut.apply(o, b, self.nv, t, h, w) for o in outs]
        if mode == "sample_pixel":
            zl_tok = (self.encoder.forward_tokens(context, slice_idx, class_idx) if zl is None
                      else convstack._TokensIn.apply(zl))
            yl = self.decoder.forward_tokens(slice, zl_tok)
            ti, hi, wi = pixel
            pred = self.ch_predictor.sample_pixel_tokens(yl, b, t * h * w, (ti * h + hi) * w + wi, temp)
            if zl is None:
                zl = convstack._TokensOut.apply(zl_tok, b, zl_tok.shape[-1], t, h, w)
            return pred, zl
        raise ValueError
