"""Configuration surface of the package (same four names as `vidgen.config`)."""
from . import config as _impl

get_cfg = _impl.get_cfg
CfgNode = _impl.CfgNode
set_global_cfg = _impl.set_global_cfg
global_cfg = _impl.global_cfg

__all__ = ("get_cfg", "CfgNode", "set_global_cfg", "global_cfg")
