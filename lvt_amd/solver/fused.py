"""torch.optim-compatible optimizers whose `step()` is a fused multi-tensor HIP launch.

Same constructor arguments, param_groups / state layout (state keys `step`, `exp_avg`, `exp_avg_sq`
for Adam; `step`, `square_avg`, `momentum_buffer` for RMSprop) and update rule as torch.optim.Adam /
torch.optim.RMSprop, so checkpoints and LR schedulers are interchangeable.  Parameters that are not on a
GPU fall back to raising: there is no CPU path in this package.
"""
import ctypes as C

import torch
from torch.optim import Optimizer

from ..hip import binding as L


def _entries(items):
    arr = (L.OptEntry * len(items))()
    for i, (p, g, s0, s1, lr, wd) in enumerate(items):
        e = arr[i]
        e.p, e.g, e.s0 = p.data_ptr(), g.data_ptr(), s0.data_ptr()
        e.s1 = s1.data_ptr() if s1 is not None else None
        e.n, e.lr, e.wd = p.numel(), lr, wd
    return arr


class _FusedBase(Optimizer):
    def zero_grad(self, set_to_none=True):
        """torch semantics; gradients living in a data-parallel bucket (engine/grad_reducer.py) are always DROPPED
        (`.grad = None`: `set_to_none=False` is not honoured for them -- a zeroed view that stays attached would make autograd
        add every new gradient into it with a kernel of its own; the reducer moves fresh gradients into the bucket instead)."""
        bucketed, plain = {}, []
        for group in self.param_groups:
            for p in group["params"]:
                ref = getattr(p, "_lvt_reducer", None)
                red = ref() if ref is not None else None
                if red is not None and red.active:
                    bucketed.setdefault(id(red), (red, set()))[1].add(p)
                else:
                    plain.append(p)
        for red, ps in bucketed.values():
            red.zero_grad(only=ps)
        for p in plain:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_()
                    p.grad.zero_()

    def _collect(self, make_state):
        """-> dict step_count -> list of (p, grad, s0, s1, lr, wd)."""
        by_step = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise L.LvtError("fused optimizers need parameters on a MI355X; there is no CPU path")
                g = p.grad
                if g.is_sparse or not g.is_contiguous() or not p.is_contiguous():
                    raise L.LvtError("fused optimizers need dense contiguous parameters and gradients")
                st = self.state[p]
                if len(st) == 0:
                    make_state(st, p, group)
                st["step"] += 1
                by_step.setdefault(int(st["step"]), []).append((p, g, st, group))
        return by_step


class FusedAdam(_FusedBase):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None

        def make(st, p, group):
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        for step, items in self._collect(make).items():
            # groups may differ in betas/eps in principle; batch by (betas, eps)
            batches = {}
            for p, g, st, group in items:
                key = (group["betas"], group["eps"])
                batches.setdefault(key, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], group["lr"],
                                                    group["weight_decay"]))
            for (betas, eps), ents in batches.items():
                arr = _entries(ents)
                L.check(L.lib().lvt_adam_step(arr, len(ents), betas[0], betas[1], eps, step, L.stream_ptr()),
                        "lvt_adam_step")
        L.bump_epoch()          # parameters were rewritten through raw pointers: cached max |.| records are stale
        return loss


class FusedRMSprop(_FusedBase):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0):
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay, momentum=momentum,
                                      centered=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None

        def make(st, p, group):
            st["step"] = 0
            st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if group["momentum"] > 0:
                st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        for step, items in self._collect(make).items():
            batches = {}
            for p, g, st, group in items:
                key = (group["alpha"], group["eps"], group["momentum"])
                batches.setdefault(key, []).append((p, g, st["square_avg"], st.get("momentum_buffer"), group["lr"],
                                                    group["weight_decay"]))
            for (alpha, eps, mom), ents in batches.items():
                arr = _entries(ents)
                L.check(L.lib().lvt_rmsprop_step(arr, len(ents), alpha, eps, mom, L.stream_ptr()), "lvt_rmsprop_step")
        L.bump_epoch()
        return loss
