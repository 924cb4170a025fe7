"""Pieces shared by the meta-architectures: weight init, data staging, DP wrapping."""
import numpy as np
import torch
from torch import nn


def _normal_fill(weight, slope):
    # He-style std for a leaky-ReLU of the given slope, fan computed over every dim but the last (as the reference does)
    fan = np.prod(weight.shape[:-1])
    weight.normal_(std=1 / np.sqrt((1 + slope ** 2) * fan))      # evaluated exactly as the reference does (same float)


def _xavier_fill(weight, slope):
    nn.init.xavier_uniform_(weight)


_FILLS = {"normal": _normal_fill, "xavier_uniform": _xavier_fill}


def init_weights(module, init_type="normal", slope=0.2):
    """The reference's initialiser (ae.py:41-61 / vt.py:34-54), same visiting order and random draws: every sub-module
    whose class name contains "Conv" or "Linear" and that owns a `.weight` is filled (normal with the std above, or
    xavier-uniform) and its bias zeroed; afterwards each DIRECT child that defines `init_weights` re-initialises itself."""
    fill = _FILLS.get(init_type)
    if fill is None:
        raise ValueError("unknown INIT_TYPE %r" % (init_type,))

    def visit(sub):
        kind = type(sub).__name__
        if ("Conv" in kind or "Linear" in kind) and hasattr(sub, "weight"):
            fill(sub.weight.data, slope)
            bias = getattr(sub, "bias", None)
            if bias is not None:
                bias.data.zero_()

    module.apply(visit)
    for child in module.children():
        own = getattr(child, "init_weights", None)
        if own is not None:
            own(init_type, slope)


def stack_to_device(items, device):
    """list of per-sample numpy arrays / tensors -> one batched device tensor with a single H2D copy
    (the reference does one `torch.as_tensor(..., device)` per sample, ae.py:153, vt.py:285-292)."""
    first = items[0]
    if isinstance(first, torch.Tensor):
        want = torch.device(device)
        if first._base is not None and first.device.type == want.type and want.index in (None, first.device.index):
            # per-sample views of ONE batched device tensor, in order (data/prefetch.py DevicePrefetcher): that tensor is the batch
            base, n = first._base, len(items)
            if (base.is_contiguous() and base.shape[0] == n and tuple(base.shape[1:]) == tuple(first.shape)
                    and all(t._base is base and t.data_ptr() == base[i].data_ptr() for i, t in enumerate(items))):
                return base
        if first.device.type == "cpu":
            return torch.stack(items, dim=0).to(device, non_blocking=True)
        return torch.stack([t.to(device) for t in items], dim=0)
    return torch.from_numpy(np.stack([np.asarray(a) for a in items], axis=0)).to(device, non_blocking=True)
