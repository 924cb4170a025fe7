from abc import ABCMeta

from torch import nn


class Encoder(nn.Module, metaclass=ABCMeta):
    """Abstract base of encoders (vidgen/modeling/encoder/encoder.py:8-25)."""
