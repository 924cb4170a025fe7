"""Training losses of the auto-encoders (`PixelLoss`: L1 / L2 reconstruction term)."""
from . import loss as _impl

PixelLoss = _impl.PixelLoss

__all__ = ("PixelLoss",)
