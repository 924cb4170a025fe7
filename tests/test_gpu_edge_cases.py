"""Ragged / minimal / odd sizes through the whole HIP path vs the CPU oracle: single frame, batch sizes that
do not fill a 128-row tile, a lone clip, DSFVT with one and three samples, the first and the last slice."""
import pytest
import torch

import seeded
from conftest import rel_err
from oracle import lvt_oracle as O
from util_models import MEAN, STD, dsfvt_cfg, vqvae_seeded

pytestmark = pytest.mark.gpu
DS = dict(blocks_e=((1, 16, 16),) * 8, blocks_d=((1, 16, 16),) * 8, stride=(16, 1, 1))


@pytest.mark.parametrize("n", [1, 3, 7])
def test_vqvae_ragged_batches(n):
    from lvt_amd.utils.events import EventStorage
    seed = 200 + n
    model, enc, dec, st = vqvae_seeded(seed, scale=0.05)
    x = seeded.seeded_input("edge", (n, 3, 64, 64), seed)
    model.eval()
    with torch.no_grad():
        out = model([{"image": x[i].numpy()} for i in range(n)], mode="inference")
    rec, lat = O.vqvae_inference(enc, dec, st, x, MEAN, STD)
    got = torch.stack([o["latent"] for o in out]).cpu()
    assert int((got != lat).sum()) <= 1
    if torch.equal(got, lat):
        assert rel_err(torch.stack([o["reconstruction"] for o in out]), rec) < 2e-5
    model.train()
    with EventStorage(0):
        losses = model([{"image": x[i].numpy()} for i in range(n)], mode="supervised")
    sum(losses.values()).backward()
    ref, _, _ = O.vqvae_supervised_loss(enc, dec, st, O.normalize(x, MEAN, STD))
    assert abs(float(losses["loss_reconstruction"].detach()) - float(ref["loss_reconstruction"])) < 2e-5 * float(ref["loss_reconstruction"])
    for p in list(model.encoder.parameters()) + list(model.generator.parameters()):
        assert p.grad is not None and bool(torch.isfinite(p.grad).all())


def test_vqvae_mixed_clip_lengths_and_modes():
    """5-D clip input (B,T,...) with T=3 and the auxiliary modes of the reference contract."""
    seed = 321
    model, enc, dec, st = vqvae_seeded(seed, scale=0.05)
    x = seeded.seeded_input("clip3", (2, 3, 3, 64, 64), seed)
    model.eval()
    with torch.no_grad():
        out = model([{"image_sequence": x[i].numpy()} for i in range(2)], mode="inference")
        assert tuple(out[1]["latent"].shape) == (3, 4, 16, 16) and tuple(out[1]["reconstruction"].shape) == (3, 3, 64, 64)
        lat = model([{"image_sequence": x[i].numpy()} for i in range(2)], mode="encoder")
        assert tuple(lat.shape) == (2, 3, 4, 16, 16) and lat.dtype == torch.int64
        rec = model([{"image_sequence": x[i].numpy()} for i in range(2)], mode="encoder_decoder")
        assert tuple(rec.shape) == (2, 3, 3, 64, 64)
        xn = model.preprocess_data([{"image": x[0, 0].numpy()}])
        assert rel_err(xn, O.normalize(x[0, :1], MEAN, STD)) < 1e-6
        assert rel_err(model.encode(xn), O.vqvae_encode(enc, st, O.normalize(x[0, :1], MEAN, STD)).float()) == 0.0
    with pytest.raises(ValueError):
        model([{"image": x[0, 0].numpy()}], mode="no_such_mode")
    with pytest.raises(ValueError):
        model([{"wrong_key": x[0, 0].numpy()}], mode="inference")


@pytest.mark.parametrize("slices", [(1,), (15, 1, 8)])
def test_dsfvt_small_batches_first_and_last_slice(slices):
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    seed = 55
    model = build_model(dsfvt_cfg())
    params = seeded.seeded_params(seeded.dsfvt_shapes(), seed)
    model.model.load_state_dict(params, strict=False)
    model.train()
    data = [O.prepare_slices(seeded.seeded_codes("e%d" % i, (16, 4, 16, 16), seed), (a, 0, 0), (16, 1, 1), (7, 1, 1), 1)
            for i, a in enumerate(slices)]
    with EventStorage(0):
        loss = model(data, mode="supervised")["loss_cross_entropy"]
    loss.backward()
    ctx = torch.stack([d["context"] for d in data]); sl = torch.stack([d["slice"] for d in data])
    si = torch.stack([d["slice_idx"] for d in data]); ig = torch.stack([d["ignore_mask"] for d in data])
    with torch.no_grad():
        ref, _ = O.vt_supervised_loss(params, ctx, sl, si, ig, **DS)
    assert abs(float(loss.detach()) - float(ref)) < 2e-5 * float(ref)
    with pytest.raises(ValueError):
        model(data, mode="bogus")


def test_dsfvt_all_positions_ignored_slice0_is_never_trained():
    """Slice 0 is entirely primed (N_PRIME=1): the reference never draws it in training (dataset_mapper.py:124);
    with every target ignored the cross-entropy of the reference is NaN (0/0) -- ours is as well."""
    from lvt_amd.hip import tx
    logits = torch.zeros(256, 512, device="cuda:0")
    tgt = torch.full((1, 4, 256), -100, dtype=torch.int64, device="cuda:0")
    loss, _, cnt = tx.xent_fwd(logits, tgt[0, 0], 1024, 1, 256, -100, 1.0)
    assert float(cnt) == 0.0 and bool(torch.isnan(loss))
