"""Encoder registry and factory (reference surface: vidgen/modeling/encoder/build.py:16-28)."""
from ...utils.registry import Registry
from .._factory import component_builder

ENCODER_REGISTRY = Registry("ENCODER")


def _base():
    from .encoder import Encoder
    return Encoder


build_encoder = component_builder(ENCODER_REGISTRY, "ENCODER", "encoder", base=_base)
