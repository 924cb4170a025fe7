import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "slow: long CPU test")
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden(dict):
    """npz fixture -> dict of torch tensors (int64 / float32 / bool preserved)."""

    def __init__(self, name):
        super().__init__()
        with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
            for k in z.files:
                if k == "_meta":
                    continue
                a = z[k]
                self[k] = a if a.dtype.kind in "US" else torch.from_numpy(np.ascontiguousarray(a).reshape(a.shape))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
