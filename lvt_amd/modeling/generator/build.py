"""Generator (decoder network) registry and factory (reference surface: vidgen/modeling/generator/build.py:19-31)."""
from ...utils.registry import Registry
from .._factory import component_builder

GENERATOR_REGISTRY = Registry("GENERATOR")


def _base():
    from .generator import Generator
    return Generator


build_generator = component_builder(GENERATOR_REGISTRY, "GENERATOR", "generator", base=_base)
