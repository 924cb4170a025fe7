"""Ragged / minimal / odd sizes through the whole HIP path vs the CPU oracle: single frame, batch sizes that
do not fill a 128-row tile, a lone clip, DSFVT with one and three samples, the first and the last slice."""
import pytest
import torch

import seeded
from conftest import rel_err
from oracle import lvt_oracle as O
from util_models import MEAN, STD, dsfvt_cfg, vqvae_seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
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


@pytest.mark.parametrize("tag,stride,kernel", [("g8", (16, 1, 1), (7, 1, 1)), ("g15_dssvt", (1, 2, 2), (1, 3, 3)),
                                               ("g16_dstsvt", (4, 2, 2), (5, 3, 3))])
def test_slice_context_kernel_equals_reference_mapper(golden, tag, stride, kernel):
    """`lvt_slice_context` run ON THE DEVICE against what the reference's DatasetMapper produced (fixtures G8, G15,
    G16: contexts / slices / slice indices / ignore masks captured from the real reference), and against the
    per-sample host builder for every slice offset of the geometry."""
    import itertools
    from lvt_amd.data.dataset_mapper import prepare_slices, prepare_slices_batch
    if tag == "g8":
        g = golden("g8_mapper")
        aa = (1, 2, 5, 9, 15)
        codes = torch.stack([g["codes"]] * len(aa))
        abcs = [(a, 0, 0) for a in aa]
        want = {k: torch.stack([g["a%d_%s" % (a, k)] for a in aa]) for k in ("context", "slice", "slice_idx", "ignore_mask")}
    else:
        g = golden(tag)
        codes = g["codes"]
        abcs = [tuple(int(x) for x in g["abc"][i]) for i in range(codes.shape[0])]
        ds = [prepare_slices(codes[i].numpy(), abcs[i], stride, kernel, 1, -1) for i in range(len(abcs))]
        want = {"context": g["context"], "slice_idx": g["slice_idx"],
                "slice": torch.stack([d["slice"] for d in ds]), "ignore_mask": torch.stack([d["ignore_mask"] for d in ds])}
    ctx, sl, sidx, ign = prepare_slices_batch(codes.to(DEV), abcs, stride, kernel, 1, -1)
    assert ctx.is_cuda and ctx.dtype == torch.int64 and ign.dtype == torch.bool
    assert torch.equal(ctx.cpu(), want["context"]) and torch.equal(sl.cpu(), want["slice"])
    assert torch.equal(sidx.cpu(), want["slice_idx"]) and torch.equal(ign.cpu(), want["ignore_mask"])
    # every offset of the geometry, several clips, n_prime 3, another pad value: device kernel == host builder
    T = 16 if stride[0] != 1 else 4
    vids = torch.randint(0, 512, (3, T, 4, 16, 16), generator=torch.Generator().manual_seed(11))
    for abc in itertools.product(range(stride[0]), range(stride[1]), range(stride[2])):
        if stride == (16, 1, 1) and abc[0] % 5:
            continue
        dev = prepare_slices_batch(vids.to(DEV), [abc] * 3, stride, kernel, 3, -7)
        host = prepare_slices_batch(vids, [abc] * 3, stride, kernel, 3, -7)
        for a_, b_ in zip(dev, host):
            assert a_.shape == b_.shape and torch.equal(a_.cpu(), b_), abc


def test_device_prefetcher_feeds_the_model_from_pinned_memory():
    """data/prefetch.py on the device: batches staged in pinned buffers by the worker thread, one asynchronous copy per key, handed
    over as per-sample views that `stack_to_device` takes without a second copy; the VQ-VAE step on them equals the step on the
    same arrays passed directly (ae.py:151-168), also when the pinned slots are being reused (more batches than slots)."""
    import numpy as np
    import torch
    from lvt_amd.data.prefetch import DevicePrefetcher
    from lvt_amd.modeling.meta_arch.common import stack_to_device
    from lvt_amd.utils.events import EventStorage
    from util_models import vqvae_seeded
    rng = np.random.default_rng(3)
    loader = [[{"image": rng.random((3, 64, 64), dtype=np.float32), "video_idx": 10 * b + i} for i in range(4)] for b in range(7)]
    model, _, _, _ = vqvae_seeded(5, scale=0.05)
    model.eval()
    pf = DevicePrefetcher(loader, "cuda:0")
    n = 0
    for data, ref in zip(pf, loader):
        x = stack_to_device([d["image"] for d in data], "cuda:0")
        assert x.is_cuda and x.data_ptr() == data[0]["image"].data_ptr() and [d["video_idx"] for d in data] == [d["video_idx"] for d in ref]
        assert torch.equal(x.cpu(), torch.from_numpy(np.stack([d["image"] for d in ref])))
        with torch.no_grad():
            a = model(data, mode="inference")
            b = model(ref, mode="inference")
        assert all(torch.equal(p["latent"], q["latent"]) and torch.equal(p["reconstruction"], q["reconstruction"]) for p, q in zip(a, b))
        n += 1
    assert n == 7 and all(buf.is_pinned() for bufs in pf._pinned.values() for buf in bufs if buf is not None)
