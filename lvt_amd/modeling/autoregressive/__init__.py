"""Autoregressive priors over latent codes: registry / factory and the abstract base (the subscale video
transformer registers itself when the factory is first used)."""
from . import autoregressive as _base
from . import build as _build

AUTOREGRESSIVE_REGISTRY, build_autoregressive = _build.AUTOREGRESSIVE_REGISTRY, _build.build_autoregressive
Autoregressive = _base.Autoregressive

__all__ = ("AUTOREGRESSIVE_REGISTRY", "build_autoregressive", "Autoregressive")
