from .build import GENERATOR_REGISTRY, build_generator
from .generator import Generator
from .resdecoder import ResDecoder

__all__ = ["GENERATOR_REGISTRY", "build_generator", "Generator", "ResDecoder"]
