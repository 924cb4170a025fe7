"""Evaluator callers on the GPU: code extraction round trip (on-disk latent format), MSE and bits/dim against
the oracle."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import seeded
from oracle import lvt_oracle as O
from util_models import MEAN, STD, dsfvt_cfg, vqvae_seeded

pytestmark = pytest.mark.gpu


def test_codes_extractor_mse_and_reload(tmp_path):
    from lvt_amd.data.latents import list_latent_videos, load_video_codes
    from lvt_amd.evaluation import build_evaluator, inference_on_dataset
    model, enc, dec, st = vqvae_seeded(11, scale=0.05)
    cfg = model.cfg
    cfg.OUTPUT_DIR = str(tmp_path)
    clips = [seeded.seeded_input("ev%d" % i, (16, 3, 64, 64), 11) for i in range(3)]
    loader = [[{"image_sequence": clips[i].numpy(), "video_idx": i}] for i in range(3)]          # batch size 1, like build_test_loader
    res = inference_on_dataset(model, loader, build_evaluator(cfg, "bair_test_seq"))
    rec, lat = O.vqvae_inference(enc, dec, st, torch.cat(clips), MEAN, STD)
    ref_mse = float(F.mse_loss(rec, torch.cat(clips)))
    assert abs(res["reconstruction"]["MSE"] - ref_mse) < 1e-4 * ref_mse
    vids = list_latent_videos(os.path.join(str(tmp_path), "inference", "bair_test_seq"))
    assert len(vids) == 3 and len(vids[0][1]) == 16
    codes = load_video_codes(*vids[1])
    assert codes.dtype == np.int64 and codes.shape == (16, 4, 16, 16)
    assert int((torch.from_numpy(codes) != lat[16:32]).sum()) <= 2


def test_bits_evaluator_matches_oracle():
    from lvt_amd.evaluation import BitsEvaluator, inference_on_dataset
    from lvt_amd.modeling import build_model
    cfg = dsfvt_cfg()
    model = build_model(cfg)
    params = seeded.seeded_params(seeded.dsfvt_shapes(), 21)
    model.model.load_state_dict(params, strict=False)
    codes = seeded.seeded_codes("bits", (16, 4, 16, 16), 21)
    res = inference_on_dataset(model, [[{"image_sequence": codes}]], BitsEvaluator("prdvqvae_test", True))
    with torch.no_grad():
        lg = O.vt_logits_for_entire_video(params, codes[None], ((1, 16, 16),) * 8, ((1, 16, 16),) * 8, (16, 1, 1), (7, 1, 1))[0]
    nll = F.cross_entropy(lg.permute(1, 0, 2, 3, 4)[None], codes.transpose(0, 1)[None], reduction="none")[0]
    ref = float(nll[:, 1:].sum()) / math.log(2) / nll[:, 1:].numel()            # first frame (N_PRIME=1) ignored
    assert abs(res["likelihood"]["bits_per_dim"] - ref) < 2e-5 * ref
