"""Encoders: the registry / factory pair, the abstract base and the residual conv encoder of the VQ-VAE."""
from . import build as _build
from . import encoder as _base
from . import resencoder as _res

ENCODER_REGISTRY, build_encoder = _build.ENCODER_REGISTRY, _build.build_encoder
Encoder = _base.Encoder
ResEncoder = _res.ResEncoder

__all__ = ("ENCODER_REGISTRY", "build_encoder", "Encoder", "ResEncoder")
