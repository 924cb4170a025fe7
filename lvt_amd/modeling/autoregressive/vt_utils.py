"""Subscale ordering helpers (reference: vidgen/modeling/autoregressive/vt_utils.py:6-57,104-128).

Pure index manipulation on int64 code grids; runs on whatever device the grid lives on (CPU inside
data-loader workers, as in the reference, or the GPU for the batched builder).  Written with strided
slicing instead of the reference's python triple loops."""
import torch
import torch.nn.functional as F


def subscale_order(st, sh, sw):
    """(idx2abc, abc2idx): raster order over the (a, b, c) slice offsets."""
    idx2abc = [(a, b, c) for a in range(st) for b in range(sh) for c in range(sw)]
    return idx2abc, {abc: i for i, abc in enumerate(idx2abc)}


def slice_mask(a, b, c, st, sh, sw, T, H, W, device=torch.device("cpu"), dtype=torch.float):
    """(1,1,T,H,W) mask of the positions congruent to (a,b,c) modulo the strides."""
    m = torch.zeros(1, 1, T, H, W, device=device, dtype=dtype)
    m[:, :, a::st, b::sh, c::sw] = 1
    return m


def visible_abc_mask(a, b, c, st, sh, sw, T, H, W, device=torch.device("cpu"), dtype=torch.float):
    """(1,1,T,H,W) mask of every slice generated strictly before (a,b,c)."""
    idx2abc, abc2idx = subscale_order(st, sh, sw)
    m = torch.zeros(1, 1, T, H, W, device=device, dtype=torch.int32)
    for (ai, bi, ci) in idx2abc[:abc2idx[(a, b, c)]]:
        m[:, :, ai::st, bi::sh, ci::sw] += 1
    return m.to(dtype)


def ss_shift(x, a, b, c, st, sh, sw, T, H, W, kt, kh, kw, pad_value=0):
    """Crop / pad `x` (.., T, H, W) so that a conv with kernel (kt,kh,kw) and stride (st,sh,sw) has its
    first window centred on the first element of slice (a,b,c)."""
    lo_hi = []
    for first, s, n, k in ((a, st, T, kt), (b, sh, H, kh), (c, sw, W, kw)):
        last = first + (n // s - 1) * s
        lo_hi.append((k // 2 - first, k // 2 - (n - last - 1)))
    (f0, f1), (h0, h1), (w0, w1) = lo_hi
    x = x[:, :, max(0, -f0):T - max(0, -f1), max(0, -h0):H - max(0, -h1), max(0, -w0):W - max(0, -w1)]
    pad = [max(0, w0), max(0, w1), max(0, h0), max(0, h1), max(0, f0), max(0, f1)]
    return F.pad(x, pad=pad, mode="constant", value=pad_value)


def slice_and_context(video, a, b, c, stride, kernel, pad_value):
    """video (B, nc, T, H, W) int64 -> (slice (B,nc,t,h,w), context shifted/padded for the encoder conv)."""
    st, sh, sw = stride
    B, nc, T, H, W = video.shape
    dev = video.device
    sl = video[:, :, a::st, b::sh, c::sw].contiguous()
    vmask = visible_abc_mask(a, b, c, st, sh, sw, T, H, W, device=dev, dtype=torch.bool)
    ctx = ss_shift(video.masked_fill(~vmask, pad_value), a, b, c, st, sh, sw, T, H, W, *kernel, pad_value=pad_value)
    return sl, ctx
