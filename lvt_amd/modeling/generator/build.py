import logging

from ...utils.registry import Registry
from .generator import Generator

GENERATOR_REGISTRY = Registry("GENERATOR")


def build_generator(cfg, **kwargs):
    """`cfg.MODEL.GENERATOR.NAME` -> instance via `from_config` (vidgen/modeling/generator/build.py:19-31)."""
    generator = GENERATOR_REGISTRY.get(cfg.MODEL.GENERATOR.NAME).from_config(cfg, **kwargs)
    assert isinstance(generator, Generator)
    logging.getLogger(__name__).info(
        "#params in generator: {}M".format(sum(p.numel() for p in generator.parameters()) / 1e6))
    return generator
