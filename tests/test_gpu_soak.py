"""End-to-end training on the product path: the models have to LEARN a fixed synthetic batch (VQ-VAE reconstruction,
DSFVT next-code prediction), with finite losses and flat device memory over a few hundred optimizer steps."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_vqvae_learns_a_fixed_batch():
    from lvt_amd.config import get_cfg
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    dev = "cuda:0"
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs/vqvae/PR-DVQVAE2.yaml"))
    cfg.MODEL.DEVICE = dev
    cfg.OUTPUT_DIR = "/tmp/lvt_soak_out"
    torch.manual_seed(1234)
    model = build_model(cfg)
    model.train()
    optimizers, _ = model.configure_optimizers_and_checkpointers()

    def vqvae_step(i):
        with EventStorage(i):
            losses = model(data, mode="supervised")
        sum(losses.values()).backward()
        for o in optimizers:
            o["optimizer"].step()
        for o in optimizers:
            o["optimizer"].zero_grad()
        return losses
    rng = np.random.RandomState(0)
    base = rng.rand(8, 1, 3, 8, 8).astype(np.float32)                      # low-frequency (learnable) clips
    clips = np.repeat(np.repeat(np.repeat(base, 16, 1), 8, 3), 8, 4)       # (8, 16, 3, 64, 64) in [0, 1]
    data = [{"image_sequence": clips[i]} for i in range(8)]
    hist, peaks = [], []
    for i in range(251):
        losses = vqvae_step(i)
        if i % 50 == 0:
            hist.append({k: float(v.detach()) for k, v in losses.items()})
            peaks.append(torch.cuda.max_memory_allocated())
    assert all(np.isfinite(list(h.values())).all() for h in hist)
    assert hist[-1]["loss_reconstruction"] < 0.1 * hist[0]["loss_reconstruction"], hist
    assert peaks[-1] == peaks[1], "device memory grew between step 50 and step 250"


def test_dsfvt_learns_a_fixed_batch():
    from lvt_amd.config import get_cfg
    from lvt_amd.data.dataset_mapper import prepare_slices_batch
    from lvt_amd.modeling import build_model
    from lvt_amd.utils.events import EventStorage
    dev = "cuda:0"
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs/vt/DSFVT.yaml"))
    cfg.MODEL.DEVICE = dev
    torch.manual_seed(7)
    model = build_model(cfg)
    model.train()
    optimizers, _ = model.configure_optimizers_and_checkpointers()
    v = cfg.MODEL.AUTOREGRESSIVE.VT
    g = torch.Generator(device="cpu").manual_seed(4321)
    codes = torch.randint(0, v.NV, (8, 16, v.NC, 16, 16), generator=g).to(dev)
    abcs = [(int(a), 0, 0) for a in torch.randint(v.N_PRIME, 16, (8,), generator=g)]
    ctx, sl, sidx, ign = prepare_slices_batch(codes, abcs, v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE)
    hist, peaks = [], []
    for i in range(151):
        with EventStorage(i):
            loss = model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
        loss.backward()
        model.finish_gradient_sync()
        for o in optimizers:
            o["optimizer"].step()
        for o in optimizers:
            o["optimizer"].zero_grad()
        if i % 50 == 0:
            hist.append(float(loss.detach()))
            peaks.append(torch.cuda.max_memory_allocated())
    assert np.isfinite(hist).all()
    assert abs(hist[0] - np.log(v.NV)) < 0.5, hist          # starts at ~ln(512): uniform prediction
    assert hist[-1] < 0.25 * hist[0], hist                  # memorises the batch
    assert peaks[-1] == peaks[1]
