"""Differentiable cross-rank sum (reference: vidgen/layers/batch_norm.py:148-160).

The reference implements the forward as all_gather + stack + sum and the backward as all_reduce.
Here both directions are a single in-place RCCL all-reduce (bit-identical sum on every rank)."""
import torch
import torch.distributed as dist

from ..utils import comm


def all_reduce_sum_(t, group=None):
    """In-place sum over ranks; no-op when not distributed."""
    if comm.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_sum_async_(t, group=None):
    """Start the in-place sum over ranks and return a handle whose .wait() orders the CURRENT stream behind it (RCCL runs
    the collective on its own stream, behind everything already enqueued on the current one); None when not distributed."""
    if comm.get_world_size() > 1:
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


class AllReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input):
        out = input.clone()
        return all_reduce_sum_(out)

    @staticmethod
    def backward(ctx, grad_output):
        g = grad_output.clone()
        return all_reduce_sum_(g)
