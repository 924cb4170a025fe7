from .build import AUTOREGRESSIVE_REGISTRY, build_autoregressive
from .autoregressive import Autoregressive

__all__ = ["AUTOREGRESSIVE_REGISTRY", "build_autoregressive", "Autoregressive"]
