"""Shared machinery of the conv encoder / decoder modules: the autograd node that runs a whole conv
stack on the HIP engine, and the boundary layout conversion."""
import torch
from torch import nn

from ..hip import binding as L
from ..hip import convnet, ew


class _ConvStackFn(torch.autograd.Function):
    """forward(x_cl, layers, *params) -> y_cl.  One autograd node per sub-network: the backward runs
    the hand-scheduled chain of lvt_amd.hip.convnet.stack_backward and hands all parameter gradients
    back at once."""

    @staticmethod
    def forward(ctx, x, layers, layers_grad, *flat):
        # layers_grad: torch.is_grad_enabled() at the CALL site (inside forward autograd has already switched it off, and
        # needs_input_grad is True for parameters even under torch.no_grad()): eval passes skip the backward weight layouts
        params = [(flat[2 * i], flat[2 * i + 1]) for i in range(len(layers))]
        with torch.no_grad():
            outs, saved = convnet.stack_forward(layers, x, params, want_grad=layers_grad and any(ctx.needs_input_grad))
        ctx.layers, ctx.saved, ctx.outs, ctx.x = layers, saved, outs, x
        ctx.shapes = [tuple(p.shape) for p in flat]
        return outs[-1]

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        with torch.no_grad():
            gx, grads = convnet.stack_backward(ctx.layers, ctx.x, ctx.outs, ctx.saved, gy,
                                               need_input_grad=ctx.needs_input_grad[0])
        flat = []
        for i, (dw, db) in enumerate(grads):
            flat.append(dw.view(ctx.shapes[2 * i]))
            flat.append(db.contiguous().view(ctx.shapes[2 * i + 1]))
        ctx.outs = ctx.saved = None
        return (gx, None, None) + tuple(flat)


def run_stack(x_cl, layers, params):
    flat = [t for wb in params for t in wb]
    return _ConvStackFn.apply(x_cl, layers, torch.is_grad_enabled(), *flat)


def nchw_to_cl(x, cpad=None):
    """(N,C,H,W) -> (N,1,H,W,Cp) channels-last, channels zero-padded to a multiple of 4."""
    L.require(x)
    n, c, h, w = x.shape
    cp = cpad or (c + 3) // 4 * 4
    return ew.to_channels_last(x.reshape(n, c, h * w), cp).view(n, 1, h, w, cp)


def cl_to_nchw(y, c):
    """(N,1,H,W,Cp) -> (N,C,H,W)."""
    n, _, h, w, cp = y.shape
    return ew.to_channels_first(y.view(n, h * w, cp), c).view(n, c, h, w)


class _LayoutIn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.c = x.shape[1]
        return nchw_to_cl(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return cl_to_nchw(g.contiguous(), ctx.c)


class _LayoutOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, c):
        ctx.cp = y.shape[-1]
        return cl_to_nchw(y, c)

    @staticmethod
    def backward(ctx, g):
        return nchw_to_cl(g.contiguous(), ctx.cp), None


def check_norm(norm, spectral):
    if norm not in ("", None) or spectral:
        raise NotImplementedError("lvt_amd implements the shipped configurations only (NORM '', no spectral "
                                  "norm); got norm=%r spectral=%r" % (norm, spectral))


class ResBlock(nn.Module):
    """Parameter container with the reference's key names (`block.1`, `block.3`);
    resencoder.py:10-21 / resdecoder.py:10-21.  Compute happens in the owning stack."""

    def __init__(self, dim, dim_res):
        super().__init__()
        self.block = nn.Sequential(nn.ReLU(True), nn.Conv2d(dim, dim_res, 3, 1, 1), nn.ReLU(True),
                                   nn.Conv2d(dim_res, dim, 1))


class _TokensIn(torch.autograd.Function):
    """(B, C, T, H, W) -> token-major (B*T*H*W, C)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        B, C = x.shape[:2]
        R = x[0, 0].numel()
        return ew.to_channels_last(x.contiguous().view(B, C, R), C).view(B * R, C)

    @staticmethod
    def backward(ctx, g):
        B, C = ctx.shape[:2]
        R = g.shape[0] // B
        return ew.to_channels_first(g.contiguous().view(B, R, C), C).view(ctx.shape)


class _TokensOut(torch.autograd.Function):
    """token-major (B*R, C) -> (B, C, T, H, W)."""

    @staticmethod
    def forward(ctx, tok, B, C, T, H, W):
        return ew.to_channels_first(tok.contiguous().view(B, T * H * W, tok.shape[-1]), C).view(B, C, T, H, W)

    @staticmethod
    def backward(ctx, g):
        B, C = g.shape[:2]
        R = g[0, 0].numel()
        return ew.to_channels_last(g.contiguous().view(B, C, R), C).view(B * R, C), None, None, None, None, None
