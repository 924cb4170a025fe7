from .build import META_ARCH_REGISTRY, build_model  # isort:skip
from .ae import AutoEncoderModel
from .vqvae import VQVAEModel
from .vt import VideoTransformerModel

__all__ = ["META_ARCH_REGISTRY", "build_model", "AutoEncoderModel", "VQVAEModel", "VideoTransformerModel"]
