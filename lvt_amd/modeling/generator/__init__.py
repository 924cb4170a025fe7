"""Generators (decoder networks): registry / factory, abstract base, residual ConvTranspose decoder of the VQ-VAE."""
from . import build as _build
from . import generator as _base
from . import resdecoder as _res

GENERATOR_REGISTRY, build_generator = _build.GENERATOR_REGISTRY, _build.build_generator
Generator = _base.Generator
ResDecoder = _res.ResDecoder

__all__ = ("GENERATOR_REGISTRY", "build_generator", "Generator", "ResDecoder")
