from .loss import PixelLoss

__all__ = ["PixelLoss"]
