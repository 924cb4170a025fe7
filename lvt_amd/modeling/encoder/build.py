import logging

from ...utils.registry import Registry
from .encoder import Encoder

ENCODER_REGISTRY = Registry("ENCODER")


def build_encoder(cfg, **kwargs):
    """`cfg.MODEL.ENCODER.NAME` -> instance via `from_config` (vidgen/modeling/encoder/build.py:16-28)."""
    encoder = ENCODER_REGISTRY.get(cfg.MODEL.ENCODER.NAME).from_config(cfg, **kwargs)
    assert isinstance(encoder, Encoder)
    logging.getLogger(__name__).info(
        "#params in encoder: {}M".format(sum(p.numel() for p in encoder.parameters()) / 1e6))
    return encoder
