"""Fused multi-tensor Adam / RMSprop vs torch.optim on the GPU (same gradients, 5 steps)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(70, 33), (5,), (3, 4, 5, 6), (100000,), (1,)] + [(17, 3)] * 70      # > 64 tensors, ragged sizes
    return [torch.randn(*s, generator=g) for s in shapes]


@pytest.mark.parametrize("kind", ["adam", "rmsprop", "rmsprop_nomom"])
def test_fused_matches_torch(kind):
    from lvt_amd.solver.fused import FusedAdam, FusedRMSprop
    dev = "cuda:0"
    a = [torch.nn.Parameter(p.clone().to(dev)) for p in _params(0)]
    b = [torch.nn.Parameter(p.clone().to(dev)) for p in _params(0)]
    groups = lambda ps: [{"params": [p], "lr": 1e-2 * (1 + i % 3), "weight_decay": 0.0 if i % 2 else 1e-3}
                         for i, p in enumerate(ps)]      # noqa: E731  (one group per parameter, like the reference)
    if kind == "adam":
        oa, ob = FusedAdam(groups(a), 1e-2, betas=(0.9, 0.9)), torch.optim.Adam(groups(b), 1e-2, betas=(0.9, 0.9))
    else:
        mom = 0.9 if kind == "rmsprop" else 0.0
        oa = FusedRMSprop(groups(a), 1e-2, alpha=0.95, momentum=mom)
        ob = torch.optim.RMSprop(groups(b), 1e-2, alpha=0.95, momentum=mom)
    for step in range(5):
        for i, (pa, pb) in enumerate(zip(a, b)):
            g = torch.randn(pa.shape, generator=torch.Generator().manual_seed(100 * step + i)).to(dev)
            if i == 4 and step % 2:          # a parameter that sometimes receives no gradient
                pa.grad = pb.grad = None
                continue
            pa.grad, pb.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    for pa, pb in zip(a, b):
        assert rel_err(pa, pb) < 2e-6
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    assert set(sa[0].keys()) == set(sb[0].keys())
