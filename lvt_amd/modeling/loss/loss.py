"""Pixel loss = lambda * (mse | l1) (reference: vidgen/modeling/loss/loss.py:5-20; the shipped configs use "l2")."""
import torch
from torch import nn

from ...hip import ew


class _MseFn(torch.autograd.Function):
    """scale * mean((a - b)^2) over `denom` real elements (pads contribute exact zeros), fixed-order
    reduction; backward w.r.t. `a` only (targets are data / detached in every call site)."""

    @staticmethod
    def forward(ctx, a, b, denom, scale, l1=False):
        ctx.save_for_backward(a, b)
        ctx.denom, ctx.scale, ctx.l1 = denom, scale, l1
        return ew.mse_fwd(a, b, denom, scale, l1=l1)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ew.mse_bwd(a, b, ctx.denom, ctx.scale, gout=g.contiguous().view(1), l1=ctx.l1), None, None, None, None


def mse(a, b, denom=None, scale=1.0):
    return _MseFn.apply(a.contiguous(), b.contiguous(), float(denom if denom is not None else a.numel()), scale)


def l1(a, b, denom=None, scale=1.0):
    """scale * mean |a - b| (F.l1_loss, loss.py:11-12)."""
    return _MseFn.apply(a.contiguous(), b.contiguous(), float(denom if denom is not None else a.numel()), scale, True)


class PixelLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        mode = cfg.LOSS.PIXEL.MODE
        if mode not in ("l1", "l2"):
            raise NotImplementedError                      # (loss.py:15-16)
        self._fn = l1 if mode == "l1" else mse
        self._lambda = cfg.LOSS.PIXEL.LAMBDA

    def forward(self, input, target, denom=None):
        return self._fn(input, target, denom, self._lambda)
