from abc import ABCMeta

from torch import nn


class Generator(nn.Module, metaclass=ABCMeta):
    """Abstract base of generators / decoders (vidgen/modeling/generator/generator.py:8-25)."""
