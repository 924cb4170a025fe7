"""Optimizer and learning-rate-schedule factories (fused multi-tensor Adam / RMSprop on the device)."""
from . import build as _build

build_optimizer, build_lr_scheduler = _build.build_optimizer, _build.build_lr_scheduler

__all__ = ("build_optimizer", "build_lr_scheduler")
