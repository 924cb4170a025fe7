from abc import ABCMeta

from torch import nn


class Autoregressive(nn.Module, metaclass=ABCMeta):
    """Abstract base of autoregressive models (vidgen/modeling/autoregressive/autoregressive.py:8-25)."""
