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


def test_vt_sampler_writes_decoded_samples(tmp_path):
    """TEST.EVALUATORS "VTSampler" (vt_sampler.py:18-81, tools/train_net.py:52-53): the transformer's inference output carries
    `samples`, the evaluator decodes them with the configured VQ-VAE and writes codes.npy + one PNG per frame per sample; the
    saved frames equal a direct decode of the saved codes."""
    from PIL import Image
    from lvt_amd.evaluation import VTSampler, build_evaluator, inference_on_dataset
    from lvt_amd.modeling import build_model
    from util_models import ROOT
    cfg = dsfvt_cfg()
    cfg.OUTPUT_DIR = str(tmp_path)
    cfg.TEST.EVALUATORS = "VTSampler"
    cfg.TEST.VT_SAMPLER.NUM_SAMPLES = 2
    cfg.TEST.VT_SAMPLER.N_PRIME = 14                              # two frames to sample: 512 positions
    cfg.TEST.VT_SAMPLER.VQ_VAE.CFG = os.path.join(ROOT, "configs/vqvae/PR-DVQVAE2.yaml")
    with pytest.raises(FileNotFoundError):                       # the configured checkpoints are not in the tree: an error, as in the reference
        build_evaluator(cfg, "prdvqvae_test")
    vs = cfg.TEST.VT_SAMPLER.VQ_VAE
    vs.ENCODER_WEIGHTS = vs.GENERATOR_WEIGHTS = vs.CODEBOOK_WEIGHTS = ""         # initialised weights
    model = build_model(cfg)
    model.model.load_state_dict(seeded.seeded_params(seeded.dsfvt_shapes(), 21), strict=False)
    ev = build_evaluator(cfg, "prdvqvae_test")
    assert isinstance(ev, VTSampler)
    codes = seeded.seeded_codes("vts", (16, 4, 16, 16), 21)
    inference_on_dataset(model, [[{"image_sequence": codes, "video_idx": 7}]], ev)
    root = os.path.join(str(tmp_path), "inference", "samples", "prdvqvae_test")
    assert sorted(os.listdir(root)) == ["video_0_7", "video_1_7"]
    for s_ in range(2):
        d = os.path.join(root, "video_%d_7" % s_)
        saved = np.load(os.path.join(d, "codes.npy"))
        assert saved.shape == (16, 4, 16, 16) and saved.dtype == np.int64
        assert np.array_equal(saved[:14], codes.numpy()[:14])                      # the priming frames are kept
        assert sorted(os.listdir(d)) == sorted(["codes.npy"] + ["%d.png" % i for i in range(16)])
        with torch.no_grad():
            fr = ev.vqvae.back_normalizer(ev.vqvae.decode(torch.from_numpy(saved).to("cuda:0"))) * 255
        fr = fr.clamp(0, 255).permute(0, 2, 3, 1).cpu().numpy().astype(np.uint8)
        png = np.asarray(Image.open(os.path.join(d, "9.png")))
        assert png.shape == (64, 64, 3) and np.array_equal(png, fr[9])
