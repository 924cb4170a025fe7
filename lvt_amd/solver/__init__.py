from .build import build_lr_scheduler, build_optimizer

__all__ = ["build_lr_scheduler", "build_optimizer"]
