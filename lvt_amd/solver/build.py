"""Optimizer / LR-scheduler factories with the reference's semantics
(vidgen/solver/build.py:12-105): ONE param group per parameter, weight decay chosen by module type /
parameter name, Adam(beta1, beta2 from SOLVER.ADAM.*) or RMSprop(alpha, momentum), and the
Identity / WarmupMultiStepLR / WarmupCosineLR schedules.  On the GPU the update is a fused multi-tensor HIP launch (solver/fused.py) with torch.optim's exact rule."""
import math
from bisect import bisect_right

import torch
from torch.optim.lr_scheduler import LambdaLR

from .fused import FusedAdam, FusedRMSprop

_NORM_TYPES = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm,
               torch.nn.GroupNorm, torch.nn.InstanceNorm1d, torch.nn.InstanceNorm2d, torch.nn.InstanceNorm3d,
               torch.nn.LayerNorm, torch.nn.LocalResponseNorm)


def _param_groups(model, lr, wd, wd_norm, wd_bias):
    groups, seen = [], set()
    for module in model.modules():
        for key, value in module.named_parameters(recurse=False):
            if not value.requires_grad or value in seen:
                continue
            seen.add(value)
            decay = wd_norm if isinstance(module, _NORM_TYPES) else (wd_bias if key == "bias" else wd)
            groups.append({"params": [value], "lr": lr, "weight_decay": decay})
    return groups


def build_optimizer(model, cfg, suffix=""):
    s = cfg.SOLVER
    lr = s["LR" + suffix]
    wd, wd_norm, wd_bias = (s.WEIGHT_DECAY["BASE" + suffix], s.WEIGHT_DECAY["NORM" + suffix],
                            s.WEIGHT_DECAY["BIAS" + suffix])
    models = model if isinstance(model, list) else [model]
    params = [g for m in models for g in _param_groups(m, lr, wd, wd_norm, wd_bias)]
    on_gpu = any(p.is_cuda for g in params for p in g["params"])
    if s.OPTIMIZER_NAME == "adam":
        cls = FusedAdam if on_gpu else torch.optim.Adam          # same rule; the fused one needs device tensors
        return cls(params, lr, betas=(s.ADAM["BETA1" + suffix], s.ADAM["BETA2" + suffix]), weight_decay=wd)
    if s.OPTIMIZER_NAME == "rmsprop":
        cls = FusedRMSprop if on_gpu else torch.optim.RMSprop
        return cls(params, lr, alpha=s.RMSPROP["ALPHA" + suffix], weight_decay=wd,
                   momentum=s.RMSPROP["MOMENTUM" + suffix])
    raise ValueError("Unknown optimizer: {}".format(s.OPTIMIZER_NAME))


def _warmup_factor(method, it, warmup_iters, warmup_factor):
    if it >= warmup_iters:
        return 1.0
    if method == "constant":
        return warmup_factor
    if method == "linear":
        alpha = it / warmup_iters
        return warmup_factor * (1 - alpha) + alpha
    raise ValueError("Unknown warmup method: {}".format(method))


def build_lr_scheduler(cfg, optimizer):
    s = cfg.SOLVER
    name = s.LR_SCHEDULER_NAME
    n = len(optimizer.param_groups)
    if name == "Identity":
        return LambdaLR(optimizer, lr_lambda=[lambda it: 1.0] * n)
    if name == "WarmupMultiStepLR":
        steps = sorted(s.STEPS)

        def f(it):
            return _warmup_factor(s.WARMUP_METHOD, it, s.WARMUP_ITERS, s.WARMUP_FACTOR) * \
                s.GAMMA ** bisect_right(steps, it)
        return LambdaLR(optimizer, lr_lambda=[f] * n)
    if name == "WarmupCosineLR":
        def f(it):
            return _warmup_factor(s.WARMUP_METHOD, it, s.WARMUP_ITERS, s.WARMUP_FACTOR) * \
                0.5 * (1.0 + math.cos(math.pi * it / s.MAX_ITER))
        return LambdaLR(optimizer, lr_lambda=[f] * n)
    raise ValueError("Unknown LR scheduler: {}".format(name))
