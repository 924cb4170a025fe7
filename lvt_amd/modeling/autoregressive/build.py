"""Autoregressive-prior registry and factory (reference surface: vidgen/modeling/autoregressive/build.py:16-29)."""
from ...utils.registry import Registry
from .._factory import component_builder

AUTOREGRESSIVE_REGISTRY = Registry("AUTOREGRESSIVE")


def _base():
    from .autoregressive import Autoregressive
    return Autoregressive


def _register_implementations():
    from . import videotransformer  # noqa: F401  (registers VideoTransformer)


build_autoregressive = component_builder(AUTOREGRESSIVE_REGISTRY, "AUTOREGRESSIVE", "autoregressive", base=_base,
                                         preload=_register_implementations)
