"""Pieces shared by the meta-architectures: weight init, data staging, DP wrapping."""
import numpy as np
import torch
from torch import nn
from torch.nn import init


def init_weights(module, init_type="normal", slope=0.2):
    """Reference initialiser (ae.py:41-61 / vt.py:34-54): every sub-module whose class name contains
    "Conv" or "Linear" and has a `.weight` gets normal(std) / xavier_uniform, bias <- 0; then direct
    children exposing `init_weights` are asked to re-initialise themselves."""

    def init_func(m):
        name = m.__class__.__name__
        if hasattr(m, "weight") and ("Conv" in name or "Linear" in name):
            if init_type == "normal":
                std = 1 / np.sqrt((1 + slope ** 2) * np.prod(m.weight.data.shape[:-1]))
                m.weight.data.normal_(std=std)
            elif init_type == "xavier_uniform":
                nn.init.xavier_uniform_(m.weight.data)
            else:
                raise ValueError
            if getattr(m, "bias", None) is not None:
                init.constant_(m.bias.data, 0.0)

    module.apply(init_func)
    for m in module.children():
        if hasattr(m, "init_weights"):
            m.init_weights(init_type, slope)


def stack_to_device(items, device):
    """list of per-sample numpy arrays / tensors -> one batched device tensor with a single H2D copy
    (the reference does one `torch.as_tensor(..., device)` per sample, ae.py:153, vt.py:285-292)."""
    first = items[0]
    if isinstance(first, torch.Tensor):
        if first.device.type == "cpu":
            return torch.stack(items, dim=0).to(device, non_blocking=True)
        return torch.stack([t.to(device) for t in items], dim=0)
    return torch.from_numpy(np.stack([np.asarray(a) for a in items], axis=0)).to(device, non_blocking=True)
