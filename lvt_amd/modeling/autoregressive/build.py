import logging

from ...utils.registry import Registry
from .autoregressive import Autoregressive

AUTOREGRESSIVE_REGISTRY = Registry("AUTOREGRESSIVE")


def build_autoregressive(cfg, **kwargs):
    """`cfg.MODEL.AUTOREGRESSIVE.NAME` -> instance (vidgen/modeling/autoregressive/build.py:16-29)."""
    from . import videotransformer  # noqa: F401  (registers VideoTransformer)
    model = AUTOREGRESSIVE_REGISTRY.get(cfg.MODEL.AUTOREGRESSIVE.NAME).from_config(cfg, **kwargs)
    assert isinstance(model, Autoregressive)
    logging.getLogger(__name__).info(
        "#params in autoregressive: {}M".format(sum(p.numel() for p in model.parameters()) / 1e6))
    return model
