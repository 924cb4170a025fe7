"""Run-to-run bit-reproducibility of whole train steps.

The library has no floating-point atomics: split-K partial sums, column sums, embedding-gradient gathers, EMA statistics and
the loss reductions are combined in fixed orders, and the max |.| scalars of the f16x2 arithmetic are folded with an INTEGER
atomic max (exact and order-independent).  So two processes-worth of identical work -- the same seed, the same batches, three
optimizer steps, here twice in one process -- must end in bit-identical losses, parameters, optimizer-visible buffers and
codebook state, in every arithmetic mode.  (The reference's CUDA path is not reproducible run to run: index_add / scatter
atomics in its embedding and EMA updates.)"""
import pytest
import torch

from util_models import dsfvt_cfg, vqvae_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(params=["f16x2", "bf16x3"])
def math_mode(request):
    from lvt_amd.hip import binding as L
    before = L.get_math_mode()
    L.set_math_mode(request.param)
    yield request.param
    L.set_math_mode(before)


def _state(model):
    out = {"p." + n: p.detach().clone() for n, p in model.named_parameters()}
    out.update({"b." + n: b.detach().clone() for n, b in model.named_buffers() if b is not None})
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def _vqvae_run(steps=3, clips_per_step=8):
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    cfg = vqvae_cfg(DEV)
    cfg.OUTPUT_DIR = "/tmp/lvt_test_out"
    torch.manual_seed(11)
    model = build_model(cfg)
    model.train()
    opts, _ = model.configure_optimizers_and_checkpointers()
    g = torch.Generator().manual_seed(5)
    losses = []
    for i in range(steps):
        clips = torch.rand(clips_per_step, 16, 3, 64, 64, generator=g).to(DEV)
        with EventStorage(i):
            ls = model([{"image_sequence": clips[j]} for j in range(clips_per_step)], mode="supervised")
        sum(ls.values()).backward()
        for o in opts:
            o["optimizer"].step()
        for o in opts:
            o["optimizer"].zero_grad()
        losses.append({k: float(v.detach()) for k, v in ls.items()})
    return losses, _state(model)


def _dsfvt_run(steps=3, slices_per_step=8):
    from lvt_amd.data.dataset_mapper import prepare_slices_batch
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    cfg = dsfvt_cfg(DEV)
    cfg.OUTPUT_DIR = "/tmp/lvt_test_out"
    torch.manual_seed(13)
    model = build_model(cfg)
    model.train()
    with torch.no_grad():
        for n, p in model.model.named_parameters():
            if n.endswith("_bank"):
                p.normal_(0, 0.2)                      # non-zero relative-position banks: their gradient path is exercised
    opts, _ = model.configure_optimizers_and_checkpointers()
    v = cfg.MODEL.AUTOREGRESSIVE.VT
    g = torch.Generator().manual_seed(7)
    losses = []
    for i in range(steps):
        codes = torch.randint(0, v.NV, (slices_per_step, 16, v.NC, 16, 16), generator=g).to(DEV)
        abcs = [(int(a), 0, 0) for a in torch.randint(v.N_PRIME, 16, (slices_per_step,), generator=g)]
        ctx, sl, sidx, ign = prepare_slices_batch(codes, abcs, v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE)
        with EventStorage(i):
            loss = model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
        loss.backward()
        for o in opts:
            o["optimizer"].step()
        for o in opts:
            o["optimizer"].zero_grad()
        losses.append(float(loss.detach()))
    return losses, _state(model)


def test_vqvae_train_steps_are_bit_reproducible(math_mode):
    l0, s0 = _vqvae_run()
    l1, s1 = _vqvae_run()
    assert l0 == l1
    assert l0[0] != l0[-1]                             # the steps did move the model
    _same(s0, s1)


def test_dsfvt_train_steps_are_bit_reproducible(math_mode):
    l0, s0 = _dsfvt_run()
    l1, s1 = _dsfvt_run()
    assert l0 == l1
    assert l0[0] != l0[-1]
    _same(s0, s1)
