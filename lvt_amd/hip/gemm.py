"""Thin typed wrappers over lvt_gemm_f32 / lvt_conv3d_* / lvt_colsum."""
import ctypes as C
import os

import torch

from . import binding as L


def gemm(A, B, C_out, M, N, K, ta=0, tb=0, lda=None, ldb=None, ldc=None, a_kb=0, a_skb=0, b_kb=0,
         b_skb=0, batch_outer=1, batch_inner=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), alpha=1.0, flags=0,
         bias=None, res=None, ldr=0, mask=None, ldm=0, splits=1, a_colsum=None, c_plane=0, a_also=None, b_also=None):
    """C = epi(alpha * A @ B); see include/lvt_hip.h for the addressing rules.  With EPI_PLANES `C_out` is a bf16 tensor
    that receives the exact 3-way bf16 split of the result (planes c_plane elements apart).
    a_also / b_also: a second tensor that the A / B operand of a batched launch reaches through its batch stride (address
    differences): its max |.| enters the f16x2 operand scale next to A's / B's own."""
    L.require(A, B, C_out, bias, res, mask)
    d = L.GemmDesc()
    d.M, d.N, d.K, d.ta, d.tb = M, N, K, ta, tb
    d.A, d.lda, d.a_kb, d.a_skb = A.data_ptr(), (lda if lda is not None else (K if ta == 0 else M)), a_kb, a_skb
    d.B, d.ldb, d.b_kb, d.b_skb = B.data_ptr(), (ldb if ldb is not None else (K if tb == 0 else N)), b_kb, b_skb
    d.C, d.ldc = C_out.data_ptr(), (ldc if ldc is not None else N)
    d.batch_outer, d.batch_inner = batch_outer, batch_inner
    d.sA_o, d.sA_i = sA
    d.sB_o, d.sB_i = sB
    d.sC_o, d.sC_i = sC
    d.alpha, d.flags = alpha, flags | L.math_flag()
    d.bias = bias.data_ptr() if bias is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.ldr = ldr if res is not None and ldr else d.ldc
    d.mask = mask.data_ptr() if mask is not None else None
    d.ldm = ldm if mask is not None and ldm else d.ldc
    d.splits = splits
    d.a_colsum = a_colsum.data_ptr() if a_colsum is not None else None
    d.c_plane = c_plane
    if L.f16x2():
        # operand scales of the f16x2 arithmetic; the launch reports max |C| unless it is a split-K one (weight gradients)
        d.a_amax, d.b_amax = L.amax_of(A).data_ptr(), L.amax_of(B).data_ptr()
        d.a_amax2 = L.amax_of(a_also).data_ptr() if a_also is not None else None
        d.b_amax2 = L.amax_of(b_also).data_ptr() if b_also is not None else None
        if splits <= 1 and C_out.dtype == torch.float32:
            d.c_amax = L.new_amax(C_out).data_ptr()
    lib = L.lib()
    ws, nws = None, 0
    if splits > 1:
        nws = lib.lvt_gemm_workspace_bytes(C.byref(d))
        ws = L.workspace(nws, A.device, "gemm")
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(lib.lvt_gemm_f32(C.byref(d), L.ptr(ws), nws, L.stream_ptr()), "lvt_gemm_f32")
    if t0 is not None:
        # causal attention products walk 3 of the 4 (tile, k-range) quarters of a 256-token block: count what is executed
        causal = 0.75 if flags & (L.CAUSAL_KMAX | L.CAUSAL_KMIN | L.CAUSAL_TILE) and M == 256 and N % 128 == 0 else 1.0
        L.TIMER.end("gemm_%s%s" % ("nt"[ta], "tn"[tb]), 2.0 * M * N * K * batch_outer * batch_inner * causal, t0)
    return C_out


class P2Image:
    """The P2 image (include/lvt_hip.h, csrc/gemm_p2.hip) of a (rows, K) fp32 matrix: `data` has the matrix's shape and dtype
    float32 but holds the image bytes (per row and group of 32 k: 32 fp16 hi terms, then 32 fp16 lo terms); `amax` is the device
    scalar whose power-of-two scale the image was made under -- every consumer must be handed the same scalar."""
    __slots__ = ("data", "amax")

    def __init__(self, data, amax):
        self.data, self.amax = data, amax


def p2_mode(force=None):
    """How the MODELS use the plane-fed GEMM (f16x2 arithmetic only; bit-identical to the engine in every mode):
      "off"  (default; LVT_P2 unset or 0; force=False; any other arithmetic);
      "qkv"  (LVT_P2=qkv): the packed q/k/v weights of every attention layer get a P2 image per pass and the q/k/v projection runs
             as lvt_gemm_p2_f32 with its fp32 A through registers and the weight tile by LDS-DMA.  Alone, on random operands, that
             launch goes 218 -> 198 us and the seven other data products of a layer +-1 us (tools/profile/p2_bonly.py); in the
             step the same launch takes 211 us against the engine's 207 and the free-running step 27.71-27.78 ms against 27.57-27.62;
      "full" (LVT_P2=1, or force=True): also the LayerNorm outputs and the first FFN weight as images, both operands of those two
             products by LDS-DMA: 27.52-27.63 ms -- nothing."""
    if not L.f16x2():
        return "off"
    if force is not None:
        return "full" if force else "off"
    env = os.environ.get("LVT_P2", "")
    if env == "qkv":
        return "qkv"
    return "full" if env not in ("", "0") else "off"


def p2_supported(force=None):
    return p2_mode(force) == "full"


def p2_pack(specs):
    """P2 images of many matrices in ONE launch per 64 (lvt_p2_pack_multi).  specs: (src, transpose, dst, amax[, batch, bs_src,
    bs_dst]) with src a 2-D fp32 view with unit column stride (row stride = src.stride(0)), dst a 2-D float32 view that receives
    the image (rows = src rows, or src columns when transpose; its row stride must be a multiple of 32) and amax the device
    scalar of the scale; batch > 1 repeats the entry for matrices bs_src / bs_dst floats apart."""
    arr = (L.P2PackEntry * len(specs))()
    for e, sp in zip(arr, specs):
        src, transpose, dst, amax = sp[:4]
        if src.dim() != 2 or dst.dim() != 2 or src.stride(1) != 1 or dst.stride(1) != 1 or not src.is_cuda:
            raise L.LvtError("p2_pack: 2-D device views with unit column stride expected")
        e.src, e.dst, e.rows, e.cols = src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1]
        e.ld_src, e.ld_dst, e.transpose, e.amax = src.stride(0), dst.stride(0), int(bool(transpose)), amax.data_ptr()
        e.batch, e.bs_src, e.bs_dst = sp[4:7] if len(sp) > 4 else (1, 0, 0)
    L.check(L.lib().lvt_p2_pack_multi(arr, len(specs), L.stream_ptr()), "lvt_p2_pack_multi")


def gemm_p2(A, Bimg, C_out, M, N, K, lda=None, a_kb=0, a_skb=0, ldb=None, ldc=None, batch_outer=1, batch_inner=1, sA=(0, 0),
            sB=(0, 0), sC=(0, 0), alpha=1.0, flags=0, bias=None, res=None, ldr=0, mask=None, ldm=0, out_image=None, out_bound=None):
    """C = epi(alpha * A @ B^T) with B a P2Image (rows = n, k contiguous) and A a P2Image or an fp32 tensor (lvt_gemm_p2_f32).
    out_image (a float32 tensor shaped like C_out) additionally receives the P2 image of the result under the scale of the
    device scalar out_bound, an a-priori bound of max |C|; returns (C_out, P2Image or None)."""
    a_img = isinstance(A, P2Image)
    At = A.data if a_img else A
    L.require(C_out, bias, res, mask)
    d = L.GemmP2Desc()
    d.M, d.N, d.K = M, N, K
    d.A, d.lda, d.a_planes, d.a_kb, d.a_skb = At.data_ptr(), (lda if lda is not None else K), int(a_img), a_kb, a_skb
    d.B, d.ldb = Bimg.data.data_ptr(), (ldb if ldb is not None else K)
    d.C, d.ldc = C_out.data_ptr(), (ldc if ldc is not None else N)
    d.batch_outer, d.batch_inner = batch_outer, batch_inner
    d.sA_o, d.sA_i = sA
    d.sB_o, d.sB_i = sB
    d.sC_o, d.sC_i = sC
    d.alpha, d.flags = alpha, flags | L.MATH_F16X2
    d.bias = bias.data_ptr() if bias is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.ldr = ldr if res is not None and ldr else d.ldc
    d.mask = mask.data_ptr() if mask is not None else None
    d.ldm = ldm if mask is not None and ldm else d.ldc
    d.a_amax = (A.amax if a_img else L.amax_of(A)).data_ptr()
    d.b_amax = Bimg.amax.data_ptr()
    d.c_amax = L.new_amax(C_out).data_ptr()
    img = None
    if out_image is not None:
        d.Cp, d.ldcp, d.cp_amax = out_image.data_ptr(), d.ldc, out_bound.data_ptr()
        img = P2Image(out_image, out_bound)
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(L.lib().lvt_gemm_p2_f32(C.byref(d), L.stream_ptr()), "lvt_gemm_p2_f32")
    if t0 is not None:
        L.TIMER.end("gemm_p2a" if a_img else "gemm_p2", 2.0 * M * N * K * batch_outer * batch_inner, t0)
    return C_out, img


def _smallm_splits(N, K):
    """k ranges of 256 for reductions of 1024 and more (one workgroup per 32 columns walks k at ~1.3 us per 128: a
    64 x 512 x 2048 product takes 23 us on 16 CUs unsplit, ~8 us as 8 x 16 workgroups plus the reduction launch)."""
    if K < 1024 or K % 256 != 0:
        return 1
    return K // 256


def gemm_small(A, B, C_out, M, N, K, tb=0, lda=None, ldb=None, ldc=None, batch=1, sB=0, sC=0, alpha=1.0, flags=0,
               bias=None, res=None, ldr=0, split_ws=None, pos=None, c_pos=0, r_pos=0):
    """C = epi(alpha * A @ B) for a few rows (incremental decoding: M = videos per step); see lvt_gemm_smallm_f32.
    `split_ws`: the caller's own split-K scratch (a callable bytes -> uint8/float tensor).  Callers that record
    launches into hipGraphs replayed on several streams (autoregressive/incremental.py) must pass one: the
    default workspace is shared per stream and every capture runs on the same capture stream.
    `pos` (int32 device scalar): row cursor read by the kernel, C += pos*c_pos and res += pos*r_pos elements."""
    L.require(A, B, bias, res, pos)
    # up to 512 rows stay on the decode kernels (one workgroup per 64 rows x 32 columns: 3x the workgroups of the
    # 128x128 engine tile at these shapes); the n-contiguous layout only exists for M <= 64
    if M > 512 or (M > 64 and (tb != 0 or K % 8 != 0)):
        if pos is not None or split_ws is not None:
            raise L.LvtError("gemm_small: M=%d, tb=%d, K=%d is served by the tile engine, which has neither a device-side "
                             "cursor nor caller-owned split-K scratch (decode groups are limited to 512 rows)" % (M, tb, K))
        return gemm(A, B, C_out, M, N, K, ta=0, tb=tb, lda=lda, ldb=ldb, ldc=ldc, batch_inner=batch, sB=(0, sB),
                    sC=(0, sC), alpha=alpha, flags=flags, bias=bias, res=res, ldr=ldr)
    splits = _smallm_splits(N, K) if (tb == 0 and batch == 1 and N % 4 == 0) else 1
    if splits > 1:
        lib = L.lib()
        nws = lib.lvt_gemm_smallm_splitk_workspace_bytes(M, N, splits)
        ws = L.workspace(nws, A.device, "smallm") if split_ws is None else split_ws(nws)
        if ws.numel() * ws.element_size() < nws:
            raise L.LvtError("gemm_small: split-K scratch of %d bytes, need %d" % (ws.numel() * ws.element_size(), nws))
        L.check(lib.lvt_gemm_smallm_splitk_f32(M, N, K, splits, L.ptr(A), lda if lda is not None else K, L.ptr(B),
                                               ldb if ldb is not None else K, L.ptr(C_out), ldc if ldc is not None else N,
                                               alpha, flags, L.ptr(bias), L.ptr(res),
                                               ldr if ldr else (ldc if ldc is not None else N), L.ptr(pos), c_pos, r_pos,
                                               L.ptr(ws), nws, L.stream_ptr()), "lvt_gemm_smallm_splitk_f32")
        return C_out
    L.check(L.lib().lvt_gemm_smallm_f32(M, N, K, tb, L.ptr(A), lda if lda is not None else K, L.ptr(B),
                                        ldb if ldb is not None else (K if tb == 0 else N), L.ptr(C_out),
                                        ldc if ldc is not None else N, batch, sB, sC, alpha, flags, L.ptr(bias),
                                        L.ptr(res), ldr if ldr else (ldc if ldc is not None else N), L.ptr(pos), c_pos,
                                        r_pos, L.stream_ptr()),
            "lvt_gemm_smallm_f32")
    return C_out


def gemm_small_partial(A, B, M, N, K, splits, ws):
    """Raw split-K partial tiles (splits, M, N) of A @ B^T for M <= 64 into the caller's buffer `ws` (float32,
    >= splits*M*N elements); the consumer (`splitsum_layernorm`) reduces them."""
    L.require(A, B, ws)
    lib = L.lib()
    nws = lib.lvt_gemm_smallm_splitk_workspace_bytes(M, N, splits)
    if ws.numel() * ws.element_size() < nws:
        raise L.LvtError("gemm_small_partial: buffer of %d bytes, need %d" % (ws.numel() * ws.element_size(), nws))
    L.check(lib.lvt_gemm_smallm_partial_f32(M, N, K, splits, L.ptr(A), K, L.ptr(B), K, L.ptr(ws), nws, L.stream_ptr()),
            "lvt_gemm_smallm_partial_f32")
    return ws


def splitsum_layernorm(partials, splits, rows, d, w, b, bias=None, res=None, eps=1e-5):
    """x = sum of the partial tiles (+ bias) (+ res); returns (x, LayerNorm(x) * w + b) from one launch."""
    L.require(partials, w, b, bias, res)
    x = torch.empty(rows, d, dtype=torch.float32, device=w.device)
    y = torch.empty(rows, d, dtype=torch.float32, device=w.device)
    L.check(L.lib().lvt_splitsum_layernorm_fwd(L.ptr(partials), splits, rows, d, L.ptr(bias), L.ptr(res), d, L.ptr(x), eps,
                                               L.ptr(w), L.ptr(b), L.ptr(y), L.stream_ptr()), "lvt_splitsum_layernorm_fwd")
    return x, y


def conv_geom(N, Ti, Hi, Wi, Ci, Co, kernel, stride, pad, out=None):
    """Forward-conv geometry; `pad` is the FRONT padding per dim; `out` overrides (To,Ho,Wo)
    (needed for asymmetric padding such as the causal conv)."""
    g = L.ConvGeom()
    g.N, g.Ti, g.Hi, g.Wi, g.Ci, g.Co = N, Ti, Hi, Wi, Ci, Co
    g.Kt, g.Kh, g.Kw = kernel
    g.st, g.sh, g.sw = stride
    g.pt, g.ph, g.pw = pad
    if out is None:
        out = tuple((i + 2 * p - k) // s + 1 for i, p, k, s in zip((Ti, Hi, Wi), pad, kernel, stride))
    g.To, g.Ho, g.Wo = out
    return g


def conv_flops(g):
    """Algorithmic FLOPs of one pass (fwd == bwd-data == bwd-weight): 2 * outputs * Ci * taps, counted
    over the device channel counts (the 3-channel image ends are carried as 4)."""
    return 2.0 * g.N * g.To * g.Ho * g.Wo * g.Co * g.Ci * g.Kt * g.Kh * g.Kw


def _same_amax(packed, w):
    """A re-layout (plus zero padding) of `w` has the same max |.|: hand the record on (f16x2 mode only)."""
    if L.f16x2():
        L.set_amax(packed, L.amax_of(w))
    return packed


def pack_weight(g, w, Ci_real, Co_real):
    L.require(w)
    wp = torch.empty(g.Kt * g.Kh * g.Kw, g.Ci, g.Co, dtype=torch.float32, device=w.device)
    L.check(L.lib().lvt_conv3d_pack_weight(C.byref(g), L.ptr(w), Ci_real, Co_real, L.ptr(wp), L.stream_ptr()),
            "lvt_conv3d_pack_weight")
    return _same_amax(wp, w)


def pack_weight_t(g, w, Ci_real, Co_real):
    """(taps, Co, Ci) weights of the convolution that is the backward-data pass of the stride-1 convolution `g`."""
    L.require(w)
    wt = torch.empty(g.Kt * g.Kh * g.Kw, g.Co, g.Ci, dtype=torch.float32, device=w.device)
    L.check(L.lib().lvt_conv3d_pack_weight_t(C.byref(g), L.ptr(w), Ci_real, Co_real, L.ptr(wt), L.stream_ptr()),
            "lvt_conv3d_pack_weight_t")
    return _same_amax(wt, w)


class PackBatch:
    """The weight packs of a whole convolution stack as ONE launch (lvt_conv3d_pack_weights_multi): `plain`, `t`, `phases`,
    `parity` allocate the destination and queue the pack (same layouts and bits as pack_weight*), `launch` issues them."""

    SINGLE = bool(os.environ.get("LVT_NO_PACK_BATCH"))        # A/B switch: one launch per pack, as before round 4

    # f16x2: the packs that a frame-resident launch will read also get their tiles as ready LDS images, right behind the fp32
    # pack in the same buffer (lvt_conv3d_weight_images, one more launch per stack); the conv wrappers below then pass
    # CONV_WEIGHT_IMAGE and the kernel stages the weight tiles by LDS-DMA.  LVT_NO_WEIGHT_IMAGES=1: the in-kernel split (A/B switch).
    IMAGES = not os.environ.get("LVT_NO_WEIGHT_IMAGES")

    def __init__(self):
        self.entries, self.keep, self.images = [], [], []

    def _add(self, kind, g, w, Ci_real, Co_real, shape, taps, image=False):
        if self.SINGLE:
            return (pack_weight, pack_weight_t, pack_weight_phases, pack_weight_parity)[kind](g, w, Ci_real, Co_real)
        L.require(w)
        n = 1
        for d in shape:
            n *= d
        cols = shape[-1]
        rows = n // cols
        extra = L.lib().lvt_conv3d_weight_image_bytes(rows, cols) if (image and self.IMAGES and L.f16x2()) else 0
        if extra:
            buf = torch.empty(n + extra // 4, dtype=torch.float32, device=w.device)
            dst = buf[:n].view(*shape)
        else:
            dst = torch.empty(*shape, dtype=torch.float32, device=w.device)
        e = L.PackEntry()
        e.w, e.dst, e.kind, e.taps, e.Ci, e.Co, e.Ci_real, e.Co_real = w.data_ptr(), dst.data_ptr(), kind, taps, g.Ci, g.Co, Ci_real, Co_real
        self.entries.append(e)
        self.keep.append(w)
        _same_amax(dst, w)
        if extra:
            ie = L.WeightImageEntry()
            ie.wp, ie.rows, ie.cols, ie.amax = dst.data_ptr(), rows, cols, L.amax_of(dst).data_ptr()
            self.images.append(ie)
            self.keep.append(dst)
            dst._lvt_wimg = True
        return dst

    def plain(self, g, w, Ci_real, Co_real):
        taps = g.Kt * g.Kh * g.Kw
        return self._add(0, g, w, Ci_real, Co_real, (taps, g.Ci, g.Co), taps, image=uses_patch_kernel(g))

    def t(self, g, w, Ci_real, Co_real):
        if (g.st, g.sh, g.sw) != (1, 1, 1):
            raise L.LvtError("pack_weight_t: stride-1 convolutions only")
        taps = g.Kt * g.Kh * g.Kw
        return self._add(1, g, w, Ci_real, Co_real, (taps, g.Co, g.Ci), taps, image=True)      # (asked for by bwd_data_as_conv)

    def phases(self, g, w, Ci_real, Co_real):
        return self._add(2, g, w, Ci_real, Co_real, (4, 4, g.Co, g.Ci), self._taps16(g), image=True)

    def parity(self, g, w, Ci_real, Co_real):
        return self._add(3, g, w, Ci_real, Co_real, (4, 4, g.Ci, g.Co), self._taps16(g), image=True)

    @staticmethod
    def _taps16(g):
        if (g.Kt, g.Kh, g.Kw) != (1, 4, 4):
            raise L.LvtError("pack_weight_phases / _parity: 4x4 kernels only")
        return 16

    def launch(self):
        if self.entries:
            arr = (L.PackEntry * len(self.entries))(*self.entries)
            L.check(L.lib().lvt_conv3d_pack_weights_multi(arr, len(self.entries), L.stream_ptr()), "lvt_conv3d_pack_weights_multi")
        if self.images:
            arr = (L.WeightImageEntry * len(self.images))(*self.images)
            L.check(L.lib().lvt_conv3d_weight_images(arr, len(self.images), L.stream_ptr()), "lvt_conv3d_weight_images")
        self.entries, self.keep, self.images = [], [], []


def uses_patch_kernel(g):
    """True when the forward pass of `g` runs on the frame-resident kernel (3x3 / pad 1 on 16x16 frames)."""
    return bool(L.lib().lvt_conv3d_uses_patch_kernel(C.byref(g), L.math_flag()))


def _wimg(wp):
    return L.CONV_WEIGHT_IMAGE if getattr(wp, "_lvt_wimg", False) else 0


def swapped_geom(g):
    """Geometry of the forward convolution that computes the backward-data pass of the stride-1 convolution `g`."""
    return conv_geom(g.N, g.To, g.Ho, g.Wo, g.Co, g.Ci, (g.Kt, g.Kh, g.Kw), (1, 1, 1),
                     (g.Kt - 1 - g.pt, g.Kh - 1 - g.ph, g.Kw - 1 - g.pw), out=(g.Ti, g.Hi, g.Wi))


def bwd_data_as_conv(g):
    """True when dx of `g` should run as a forward convolution over transposed weights: stride 1 and the swapped
    geometry is served by the frame-resident kernel (3x3 / pad 1 on 16x16 frames)."""
    if (g.st, g.sh, g.sw) != (1, 1, 1):
        return False
    return bool(L.lib().lvt_conv3d_uses_patch_kernel(C.byref(swapped_geom(g)), L.math_flag()))


def pack_weight_parity(g, w, Ci_real, Co_real):
    """(4 parity classes, 4 taps, Ci, Co) weights of the 4x4 / stride 2 convolution `g` for the frame-resident kernel."""
    L.require(w)
    wq = torch.empty(4, 4, g.Ci, g.Co, dtype=torch.float32, device=w.device)
    L.check(L.lib().lvt_conv3d_pack_weight_parity(C.byref(g), L.ptr(w), Ci_real, Co_real, L.ptr(wq), L.stream_ptr()),
            "lvt_conv3d_pack_weight_parity")
    return _same_amax(wq, w)


def fwd_by_parity(g):
    """True when the forward pass of `g` (4x4 / stride 2, 32x32 -> 16x16 frames) is served by the frame-resident kernel."""
    return bool(L.lib().lvt_conv3d_fwd_uses_parity_kernel(C.byref(g), L.math_flag()))


def conv_fwd(g, x, wp, bias=None, res=None, mask=None, flags=0, timer_key="conv_fwd", wq=None):
    L.require(x, wp if wq is None else wq, bias, res, mask)
    y = torch.empty(g.N, g.To, g.Ho, g.Wo, g.Co, dtype=torch.float32, device=x.device)
    if bias is not None:
        flags |= L.EPI_BIAS
    if res is not None:
        flags |= L.EPI_RESIDUAL
    if mask is not None:
        flags |= L.EPI_MASK
    io = L.amax_io(x, wp if wq is None else wq, y)
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    if wq is not None:
        L.check(L.lib().lvt_conv3d_fwd_parity(C.byref(g), L.ptr(x), L.ptr(wq), L.ptr(bias), L.ptr(res), L.ptr(mask), L.ptr(y),
                                              flags | L.math_flag() | _wimg(wq), L.io_ref(io), L.stream_ptr()), "lvt_conv3d_fwd_parity")
    else:
        L.check(L.lib().lvt_conv3d_fwd(C.byref(g), L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(res), L.ptr(mask), L.ptr(y),
                                       flags | L.math_flag() | _wimg(wp), L.io_ref(io), L.stream_ptr()), "lvt_conv3d_fwd")
    if t0 is not None:
        L.TIMER.end(timer_key, conv_flops(g), t0)
    return y


def pack_weight_phases(g, w, Ci_real, Co_real):
    """(4 phases, 4 taps, Co, Ci) weights of the stride-2 transposed pass of the 4x4 convolution `g`."""
    L.require(w)
    wph = torch.empty(4, 4, g.Co, g.Ci, dtype=torch.float32, device=w.device)
    L.check(L.lib().lvt_conv3d_pack_weight_phases(C.byref(g), L.ptr(w), Ci_real, Co_real, L.ptr(wph), L.stream_ptr()),
            "lvt_conv3d_pack_weight_phases")
    return _same_amax(wph, w)


def bwd_data_by_phases(g):
    """True when dx of `g` (4x4 / stride 2 between 32x32 and 16x16 frames) is served by the frame-resident kernel."""
    return bool(L.lib().lvt_conv3d_bwd_data_uses_phase_kernel(C.byref(g), L.math_flag()))


def conv_bwd_data(g, dy, wp, bias=None, res=None, mask=None, flags=0, wt=None, wph=None):
    """dx of the convolution `g`.  With `wt` (pack_weight_t) the pass runs as a forward convolution on the swapped
    geometry -- the frame-resident kernel for the 3x3 layers; with `wph` (pack_weight_phases) phase by phase on the same
    kernel (4x4 / stride 2 layers)."""
    if wt is not None:
        return conv_fwd(swapped_geom(g), dy, wt, bias=bias, res=res, mask=mask, flags=flags, timer_key="conv_bwd_data")
    if wph is not None:
        L.require(dy, wph, bias, res, mask)
        dx = torch.empty(g.N, g.Ti, g.Hi, g.Wi, g.Ci, dtype=torch.float32, device=dy.device)
        flags |= (L.EPI_BIAS if bias is not None else 0) | (L.EPI_RESIDUAL if res is not None else 0) | \
            (L.EPI_MASK if mask is not None else 0)
        io = L.amax_io(dy, wph, dx)
        t0 = L.TIMER.begin() if L.TIMER is not None else None
        L.check(L.lib().lvt_conv3d_bwd_data_phases(C.byref(g), L.ptr(dy), L.ptr(wph), L.ptr(bias), L.ptr(res), L.ptr(mask),
                                                   L.ptr(dx), flags | L.math_flag() | _wimg(wph), L.io_ref(io), L.stream_ptr()),
                "lvt_conv3d_bwd_data_phases")
        if t0 is not None:
            L.TIMER.end("conv_bwd_data", conv_flops(g), t0)
        return dx
    L.require(dy, wp, bias, res, mask)
    dx = torch.empty(g.N, g.Ti, g.Hi, g.Wi, g.Ci, dtype=torch.float32, device=dy.device)
    if bias is not None:
        flags |= L.EPI_BIAS
    if res is not None:
        flags |= L.EPI_RESIDUAL
    if mask is not None:
        flags |= L.EPI_MASK
    io = L.amax_io(dy, wp, dx)
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(L.lib().lvt_conv3d_bwd_data(C.byref(g), L.ptr(dy), L.ptr(wp), L.ptr(bias), L.ptr(res), L.ptr(mask),
                                        L.ptr(dx), flags | L.math_flag(), L.io_ref(io), L.stream_ptr()), "lvt_conv3d_bwd_data")
    if t0 is not None:
        L.TIMER.end("conv_bwd_data", conv_flops(g), t0)
    return dx


def convT4_fwd(x, w, bias, act_tanh):
    """ConvTranspose2d(Ci -> <=3, k4 s2 p1) forward on the dedicated image-side kernel.
    x (N,1,Hi,Wi,Ci) -> (N,1,2Hi,2Wi,4)."""
    L.require(x, w, bias)
    N, _, Hi, Wi, Ci = x.shape
    y = torch.empty(N, 1, 2 * Hi, 2 * Wi, 4, dtype=torch.float32, device=x.device)
    io = L.amax_io(x, w)
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(L.lib().lvt_convt4_fwd(L.ptr(x), L.ptr(w), L.ptr(bias), N, Hi, Wi, Ci, w.shape[1], 1 if act_tanh else 0,
                                   L.ptr(y), L.math_flag(), L.io_ref(io), L.stream_ptr()), "lvt_convt4_fwd")
    if t0 is not None:
        L.TIMER.end("thin_convT_fwd", 2.0 * N * 4 * Hi * Wi * 4 * 4 * Ci, t0)
    return y


def conv_bwd_weight(g, x, dy, Ci_real, Co_real, want_bias=False, bias_of_x=False):
    """-> dw, or (dw, db) with want_bias: db = column sums of dy (the bias gradient of a forward conv), produced by
    the same launch.  bias_of_x: db = column sums of x instead (the bias gradient of a transposed layer, whose weight
    gradient is this call with the operands swapped); db is None where the launch cannot produce it."""
    L.require(x, dy)
    lib = L.lib()
    dw = torch.empty(Co_real, Ci_real, g.Kt, g.Kh, g.Kw, dtype=torch.float32, device=x.device)
    nws = lib.lvt_conv3d_bwd_weight_workspace_bytes(C.byref(g))
    ws = L.workspace(nws, x.device, "wgrad")
    xflag = L.WGRAD_DB_OF_X if bias_of_x else 0
    fused_bias = want_bias and bool(lib.lvt_conv3d_bwd_weight_fuses_bias(C.byref(g), L.math_flag() | xflag))
    db = torch.empty(Ci_real if bias_of_x else Co_real, dtype=torch.float32, device=x.device) if fused_bias else None
    io = L.amax_io(x, dy)
    t0 = L.TIMER.begin() if L.TIMER is not None else None
    L.check(lib.lvt_conv3d_bwd_weight(C.byref(g), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), Ci_real, Co_real,
                                      L.math_flag() | (xflag if fused_bias else 0), L.io_ref(io), L.ptr(ws), nws, L.stream_ptr()),
            "lvt_conv3d_bwd_weight")
    if t0 is not None:
        L.TIMER.end("conv_bwd_weight", conv_flops(g), t0)
    return (dw, db) if want_bias else dw


def colsum(gmat, M, N, ld=None):
    L.require(gmat)
    lib = L.lib()
    out = torch.empty(N, dtype=torch.float32, device=gmat.device)
    nws = lib.lvt_colsum_workspace_bytes(M, N)
    ws = L.workspace(nws, gmat.device, "colsum")
    L.check(lib.lvt_colsum(L.ptr(gmat), M, N, ld if ld is not None else N, L.ptr(out), L.ptr(ws), nws,
                           L.stream_ptr()), "lvt_colsum")
    return out
