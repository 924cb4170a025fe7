"""BASELINE.json configs[1] / [2] at FULL size through the train path (forward + backward), in both math modes.

The oracle cannot run 512 frames or 64 slices in test time, so the full-size steps are checked through a property that
holds for the exact arithmetic and that every size-dependent choice of the kernels (split-K factors, tile schedules,
frame-resident vs implicit-GEMM convolution, bucketed reductions) has to preserve: the losses are batch means, so

    gradient of the full batch  ==  mean over K equal chunks of the chunk gradients     (same weights, same codebook)
    loss of the full batch      ==  mean of the chunk losses

The chunk runs use the SAME kernels at a quarter of the size, i.e. other split counts and grids; the 2-frame / 2-slice
runs of those kernels are pinned against the reference's goldens elsewhere (test_gpu_vqvae.py G5, test_gpu_vt.py G12).
"""
import copy

import pytest
import torch

import seeded
from conftest import rel_err
from util_models import dsfvt_cfg, vqvae_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(params=["bf16x3", "f32"])
def math_mode(request):
    from lvt_amd.hip import binding as L
    before = L.get_math_mode()
    L.set_math_mode(request.param)
    yield request.param
    L.set_math_mode(before)


def _grads(mods):
    return {"%d.%s" % (i, n): p.grad.detach().clone() for i, m in enumerate(mods) for n, p in m.named_parameters()
            if p.grad is not None}


def test_vqvae_train_step_512_frames_equals_mean_of_chunks(math_mode):
    """32 clips x 16 frames (BASELINE configs[1]).  The commitment term reads the codebook AFTER its EMA update, which
    depends on the whole batch, so it is switched off (BETA 0) for the additivity check of the gradients; the
    reconstruction path -- encoder, nearest-code search, straight-through, decoder and their backward -- is complete."""
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    cfg = vqvae_cfg(DEV)
    cfg.MODEL.CODEBOOK.BETA = 0.0
    torch.manual_seed(3)
    model = build_model(cfg)
    model.codebook.load_state_dict(seeded.seeded_codebook_state(5, scale=0.05))
    model.train()
    cb0 = copy.deepcopy(model.codebook.state_dict())
    clips = torch.rand(32, 16, 3, 64, 64, generator=torch.Generator().manual_seed(17))

    def run(sel):
        model.codebook.load_state_dict(cb0)                      # every run starts from the same codebook
        for p in model.parameters():
            p.grad = None
        with EventStorage(0):
            losses = model([{"image_sequence": clips[i].numpy()} for i in sel], mode="supervised")
        sum(losses.values()).backward()
        idx = model.codebook.last_indices.clone()
        return float(losses["loss_reconstruction"]), _grads([model.encoder, model.generator]), idx

    loss_full, g_full, idx_full = run(range(32))
    parts = [run(range(8 * c, 8 * c + 8)) for c in range(4)]
    assert abs(loss_full - sum(p[0] for p in parts) / 4) < 2e-6 * abs(loss_full)
    # the code indices of the full batch are those of the chunks, bit for bit (row-wise search, batch-size independent)
    assert torch.equal(idx_full, torch.cat([p[2] for p in parts], 0))
    worst = 0.0
    for k, gf in g_full.items():
        acc = parts[0][1][k].double()
        for p in parts[1:]:
            acc = acc + p[1][k].double()                         # fixed order
        worst = max(worst, rel_err(gf, acc / 4))
    assert worst < 1e-5, worst


def test_dsfvt_train_step_64_slices_equals_mean_of_chunks(math_mode):
    """64 slices (BASELINE configs[2]): loss and every gradient of the full batch == mean over four 16-slice chunks."""
    from lvt_amd.data.dataset_mapper import prepare_slices_batch
    from lvt_amd.modeling import build_model
    cfg = dsfvt_cfg(DEV)
    torch.manual_seed(9)
    model = build_model(cfg)
    model.train()
    # non-zero relative-position banks so that their gradients and the bias path are exercised at size
    with torch.no_grad():
        for n, p in model.model.named_parameters():
            if n.endswith("_bank"):
                p.normal_(0, 0.2)
    v = cfg.MODEL.AUTOREGRESSIVE.VT
    g = torch.Generator().manual_seed(23)
    codes = torch.randint(0, v.NV, (64, 16, v.NC, 16, 16), generator=g).to(DEV)
    abcs = [(int(a), 0, 0) for a in torch.randint(v.N_PRIME, 16, (64,), generator=g)]

    def run(lo, hi):
        for p in model.parameters():
            p.grad = None
        ctx, sl, sidx, ign = prepare_slices_batch(codes[lo:hi], abcs[lo:hi], v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE)
        loss = model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
        loss.backward()
        return float(loss), _grads([model.model])

    loss_full, g_full = run(0, 64)
    parts = [run(16 * c, 16 * c + 16) for c in range(4)]
    assert abs(loss_full - sum(p[0] for p in parts) / 4) < 2e-6 * abs(loss_full)
    assert len(g_full) > 250
    worst, worst_k = 0.0, None
    for k, gf in g_full.items():
        if k.endswith("dt_bank"):
            # a DSFVT block is one frame deep: the temporal bank has ONE column, i.e. the same number added to every
            # score of a row, whose softmax gradient is exactly zero -- what is left is rounding noise
            # (the fused backward forms dS with delta_i = sum_d dO_id O_id, rounded separately from sum_j P_ij dP_ij: the row
            # sums of dS then cancel to ~1e-7 of their terms instead of exactly, i.e. ~1e-3 of the h-bank gradient here)
            assert float(gf.abs().max()) < 5e-3 * float(g_full[k.replace("dt_bank", "dh_bank")].abs().max()), k
            continue
        acc = parts[0][1][k].double()
        for p in parts[1:]:
            acc = acc + p[1][k].double()
        e = rel_err(gf, acc / 4)
        if e > worst:
            worst, worst_k = e, k
    assert worst < 1e-5, (worst, worst_k)
