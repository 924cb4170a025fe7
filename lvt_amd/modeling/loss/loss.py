"""Pixel loss = lambda * mse (reference: vidgen/modeling/loss/loss.py:5-20; only "l2" is shipped)."""
import torch
from torch import nn

from ...hip import ew


class _MseFn(torch.autograd.Function):
    """scale * mean((a - b)^2) over `denom` real elements (pads contribute exact zeros), fixed-order
    reduction; backward w.r.t. `a` only (targets are data / detached in every call site)."""

    @staticmethod
    def forward(ctx, a, b, denom, scale):
        ctx.save_for_backward(a, b)
        ctx.denom, ctx.scale = denom, scale
        return ew.mse_fwd(a, b, denom, scale)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ew.mse_bwd(a, b, ctx.denom, ctx.scale, gout=g.contiguous().view(1)), None, None, None


def mse(a, b, denom=None, scale=1.0):
    return _MseFn.apply(a.contiguous(), b.contiguous(), float(denom if denom is not None else a.numel()), scale)


class PixelLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        mode = cfg.LOSS.PIXEL.MODE
        if mode != "l2":
            raise NotImplementedError("PixelLoss mode %r: only 'l2' is used by the shipped configs" % mode)
        self._lambda = cfg.LOSS.PIXEL.LAMBDA

    def forward(self, input, target, denom=None):
        return mse(input, target, denom, self._lambda)
