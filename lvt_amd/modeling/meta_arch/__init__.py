"""Whole models behind `build_model(cfg)`.  The registry module is imported first: the three model modules register
themselves with it at import time."""
from . import build as _build  # isort:skip
from . import ae as _ae, vqvae as _vqvae, vt as _vt

META_ARCH_REGISTRY, build_model = _build.META_ARCH_REGISTRY, _build.build_model
AutoEncoderModel, VQVAEModel, VideoTransformerModel = _ae.AutoEncoderModel, _vqvae.VQVAEModel, _vt.VideoTransformerModel

__all__ = ("META_ARCH_REGISTRY", "build_model", "AutoEncoderModel", "VQVAEModel", "VideoTransformerModel")
