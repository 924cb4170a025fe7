from .autoregressive import AUTOREGRESSIVE_REGISTRY, build_autoregressive
from .encoder import ENCODER_REGISTRY, build_encoder
from .generator import GENERATOR_REGISTRY, build_generator
from .loss import PixelLoss
from .meta_arch import META_ARCH_REGISTRY, build_model

__all__ = ["AUTOREGRESSIVE_REGISTRY", "build_autoregressive", "ENCODER_REGISTRY", "build_encoder",
           "GENERATOR_REGISTRY", "build_generator", "PixelLoss", "META_ARCH_REGISTRY", "build_model"]
