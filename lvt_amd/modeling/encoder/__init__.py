from .build import ENCODER_REGISTRY, build_encoder
from .encoder import Encoder
from .resencoder import ResEncoder

__all__ = ["ENCODER_REGISTRY", "build_encoder", "Encoder", "ResEncoder"]
