"""Helpers shared by the GPU parity tests: build lvt_amd models on cuda:0 with seeded weights."""
import os

import torch

import seeded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEAN = STD = (0.5, 0.5, 0.5)


def vqvae_cfg(device="cuda"):
    from lvt_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs/vqvae/PR-DVQVAE2.yaml"))
    cfg.MODEL.DEVICE = device
    cfg.OUTPUT_DIR = "/tmp/lvt_test_out"
    return cfg


def dsfvt_cfg(device="cuda"):
    from lvt_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs/vt/DSFVT.yaml"))
    cfg.MODEL.DEVICE = device
    cfg.OUTPUT_DIR = "/tmp/lvt_test_out"
    return cfg


def vqvae_seeded(seed, scale=None, device="cuda"):
    """(model, enc params, dec params, codebook state) with weights from tests/golden/seeded.py."""
    from lvt_amd.modeling import build_model
    model = build_model(vqvae_cfg(device))
    enc = seeded.seeded_params(seeded.VQVAE_ENCODER_SHAPES, seed, "enc.")
    dec = seeded.seeded_params(seeded.VQVAE_DECODER_SHAPES, seed, "dec.")
    model.encoder.load_state_dict(enc)
    model.generator.load_state_dict(dec)
    st = None
    if scale is not None:
        st = seeded.seeded_codebook_state(seed, scale=scale)
        model.codebook.load_state_dict(st)
    return model, enc, dec, st


def margin_ok(z_rows, codebook, rel=1e-5):
    """rows whose fp64 best/second-best squared-distance gap is large enough that every correctly
    rounded fp32 evaluation order must agree on the argmin."""
    from oracle import lvt_oracle as O
    d0, d1, _ = O.vq_margin_fp64(z_rows, codebook)
    scale = (z_rows.double() ** 2).sum(-1).reshape(-1) + (codebook.double() ** 2).sum(-1).max()
    return (d1 - d0) > rel * scale
