"""Typed wrappers over the HBM-bound helper kernels (csrc/elementwise.hip)."""
import ctypes as C

import torch

from . import binding as L


def _f32(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


def to_channels_last(x_bcr, ldo, mode=0, a=None, s=None):
    """(B, C, R) -> (B, R, ldo) with zero-padded channels; mode 1 applies (x - a[c]) / s[c]."""
    L.require(x_bcr, a, s)
    B, Cc, R = x_bcr.shape
    out = _f32(B, R, ldo, like=x_bcr)
    L.check(L.lib().lvt_to_channels_last(L.ptr(x_bcr), B, Cc, R, ldo, mode, L.ptr(a), L.ptr(s), L.ptr(out),
                                         L.out_amax(out), L.stream_ptr()), "lvt_to_channels_last")
    return out


def to_channels_first(x_brc, Cc, mode=0, a=None, s=None, lo=0.0, hi=0.0):
    """(B, R, ldi) -> (B, C, R); mode 2 applies clamp(x * s[c] + a[c], lo, hi)."""
    L.require(x_brc, a, s)
    B, R, ldi = x_brc.shape
    out = _f32(B, Cc, R, like=x_brc)
    L.check(L.lib().lvt_to_channels_first(L.ptr(x_brc), B, Cc, R, ldi, mode, L.ptr(a), L.ptr(s), lo, hi,
                                          L.ptr(out), L.stream_ptr()), "lvt_to_channels_first")
    return out


def _red_ws(dev):
    n = L.lib().lvt_reduce_workspace_bytes()
    return L.workspace(n, dev, "reduce"), n


def mse_fwd(a, b, denom, scale=1.0, l1=False):
    L.require(a, b)
    out = _f32(1, like=a)
    ws, n = _red_ws(a.device)
    L.check((L.lib().lvt_l1_fwd if l1 else L.lib().lvt_mse_fwd)(L.ptr(a), L.ptr(b), a.numel(), float(denom), scale, L.ptr(out), L.ptr(ws), n,
                                L.stream_ptr()), "lvt_mse_fwd")
    return out.view(())


def mse_bwd(a, b, denom, scale=1.0, gout=None, add=None, tanh_of_a=False, l1=False):
    L.require(a, b, gout, add)
    out = torch.empty_like(a)
    L.check((L.lib().lvt_l1_bwd if l1 else L.lib().lvt_mse_bwd)(L.ptr(a), L.ptr(b), a.numel(), float(denom), scale, L.ptr(gout), L.ptr(add),
                                1 if tanh_of_a else 0, L.ptr(out), L.out_amax(out), L.stream_ptr()), "lvt_mse_bwd")
    return out


def tanh_bwd(g, y):
    L.require(g, y)
    out = torch.empty_like(g)
    L.check(L.lib().lvt_tanh_bwd(L.ptr(g), L.ptr(y), g.numel(), L.ptr(out), L.out_amax(out), L.stream_ptr()), "lvt_tanh_bwd")
    return out


def axpy(x, alpha=1.0, alpha_dev=None, add=None):
    L.require(x, alpha_dev, add)
    out = torch.empty_like(x)
    L.check(L.lib().lvt_axpy(L.ptr(x), L.ptr(add), x.numel(), L.ptr(alpha_dev), alpha, L.ptr(out), L.stream_ptr()),
            "lvt_axpy")
    return out


def add_periodic_(x, table, P):
    L.require(x, table)
    d = x.shape[-1]
    L.check(L.lib().lvt_add_periodic(L.ptr(x), L.ptr(table), x.numel() // d, P, d, L.stream_ptr()),
            "lvt_add_periodic")
    L.drop_amax(x)          # rewritten in place behind torch's back
    return x


def layernorm_fwd(x, w, b, eps=1e-5, save_stats=True):
    L.require(x, w, b)
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x)
    mean = _f32(rows, like=x) if save_stats else None
    rstd = _f32(rows, like=x) if save_stats else None
    # f16x2: the output feeds engine launches; its max |.| is the a-priori bound max |w| sqrt(d - 1) + max |b| (one store)
    f16 = L.f16x2()
    L.check(L.lib().lvt_layernorm_fwd(L.ptr(x), rows, d, eps, L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(mean),
                                      L.ptr(rstd), L.out_amax(y), L.ptr(L.amax_of(w)) if f16 else None,
                                      L.ptr(L.amax_of(b)) if f16 else None, L.stream_ptr()), "lvt_layernorm_fwd")
    return y, mean, rstd


def layernorm_fwd_p2(x, w, b, eps=1e-5):
    """layernorm_fwd that also writes the output as a P2 image (csrc/gemm_p2.hip) under the a-priori bound it stores as the
    output's max |.|: -> (y, image data (float32-typed, y's shape), mean, rstd).  f16x2 arithmetic only."""
    L.require(x, w, b)
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x)
    yp = torch.empty_like(x)
    mean, rstd = _f32(rows, like=x), _f32(rows, like=x)
    L.check(L.lib().lvt_layernorm_fwd_p2(L.ptr(x), rows, d, eps, L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(yp), L.ptr(mean),
                                         L.ptr(rstd), L.out_amax(y), L.ptr(L.amax_of(w)), L.ptr(L.amax_of(b)),
                                         L.stream_ptr()), "lvt_layernorm_fwd_p2")
    return y, yp, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, w, add=None):
    L.require(dy, x, mean, rstd, w, add)
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(x)
    dw, db = _f32(d, like=x), _f32(d, like=x)
    n = L.lib().lvt_layernorm_bwd_workspace_bytes(d)
    ws = L.workspace(n, x.device, "ln")
    L.check(L.lib().lvt_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(w), rows, d, L.ptr(add),
                                      L.ptr(dx), L.ptr(dw), L.ptr(db), L.out_amax(dx), L.ptr(ws), n, L.stream_ptr()),
            "lvt_layernorm_bwd")
    return dx, dw, db


def row_gather(x, perm, S):
    """x (b*S, d) token matrix -> rows regrouped per sample: out[b*S + i] = x[b*S + perm[i]]."""
    L.require(x, perm)
    d = x.shape[-1]
    out = torch.empty_like(x)
    L.check(L.lib().lvt_row_gather(L.ptr(x), L.ptr(perm), x.numel() // (S * d), S, d, L.ptr(out), L.stream_ptr()),
            "lvt_row_gather")
    if L.f16x2() and L._valid_amax(x) is not None:
        L.set_amax(out, L._valid_amax(x))        # a permutation of the rows: same max |.|
    return out
