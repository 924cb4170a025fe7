"""End-to-end smoke of the drop-in drivers on the GPU: tools/train_net.py (both models, synthetic data,
checkpoints written in the reference's layout) and scripts/generate_videos.py on the five example frames."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout + r.stderr


def test_train_net_vqvae_then_dsfvt(tmp_path):
    out = str(tmp_path / "vq")
    log = _run(["tools/train_net.py", "--config-file", "configs/vqvae/PR-DVQVAE2.yaml", "--synthetic", "--max-iter", "6",
                "OUTPUT_DIR", out, "SOLVER.IMS_PER_BATCH", "8", "SOLVER.CHECKPOINT_PERIOD", "3", "SOLVER.MAX_ITER", "6"])
    assert "loss_reconstruction" in log and "loss_commitment" in log
    for sub in ("netE", "netG", "netC"):
        assert os.path.isfile(os.path.join(out, sub, "model_0000002.pth")), sub
        assert os.path.isfile(os.path.join(out, sub, "model_final.pth")), sub
    ck = torch.load(os.path.join(out, "netC", "model_final.pth"))
    assert "ve.0.embedding.weight" in ck["model"] and "ve.3.running_sum" in ck["model"]
    assert os.path.isfile(os.path.join(out, "config.yaml"))
    out2 = str(tmp_path / "vt")
    log = _run(["tools/train_net.py", "--config-file", "configs/vt/DSFVT.yaml", "--synthetic", "--max-iter", "3",
                "OUTPUT_DIR", out2, "SOLVER.IMS_PER_BATCH", "4", "SOLVER.MAX_ITER", "3"])
    assert "loss_cross_entropy" in log
    ck = torch.load(os.path.join(out2, "netG", "model_final.pth"))
    assert "decoder.block_local_attention.3.mha.w_q" in ck["model"]
    # --eval-only inference path of the VQ-VAE from the checkpoints just written
    log = _run(["tools/train_net.py", "--config-file", "configs/vqvae/PR-DVQVAE2.yaml", "--synthetic", "--eval-only",
                "OUTPUT_DIR", out, "SOLVER.IMS_PER_BATCH", "4", "MODEL.ENCODER.WEIGHTS", os.path.join(out, "netE", "model_final.pth")])
    assert "evaluation results" in log and "MSE" in log
    assert os.path.isdir(os.path.join(out, "inference"))                   # CodesExtractor wrote the latent clips
    # ... and of the transformer: BitsEvaluator through build_evaluator / inference_on_dataset (reference tools/train_net.py:35-57,76-86)
    log = _run(["tools/train_net.py", "--config-file", "configs/vt/DSFVT.yaml", "--synthetic", "--eval-only", "--eval-batches", "1",
                "OUTPUT_DIR", out2, "SOLVER.IMS_PER_BATCH", "2", "MODEL.GENERATOR.WEIGHTS", os.path.join(out2, "netG", "model_final.pth")])
    assert "bits_per_dim" in log


def test_generate_videos_on_example_frames(tmp_path, golden):
    from PIL import Image
    frames = golden("g6_inference")["frames_u8"].numpy()             # the reference's five example frames
    vdir = tmp_path / "prime"
    vdir.mkdir()
    for i, f in enumerate(frames):
        Image.fromarray(f.transpose(1, 2, 0)).save(vdir / ("%d.png" % i))
    out = str(tmp_path / "sample")
    # restrict the sampled region for the smoke test: generate 2 frames (frames 5 and 6), keep the rest primed
    log = _run(["scripts/generate_videos.py", "--video-dir", str(vdir), "--config-file", "configs/vt/DSFVT.yaml",
                "--random-weights", "OUTPUT_DIR", out], timeout=1500)
    assert "Sampled new video." in log and "Saved new video" in log
    # without the flag a configured-but-absent checkpoint is an error, not a silent run on random weights
    r = subprocess.run([sys.executable, "scripts/generate_videos.py", "--video-dir", str(vdir), "--config-file",
                        "configs/vt/DSFVT.yaml", "OUTPUT_DIR", out], cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "not found" in r.stderr
    pngs = sorted(os.listdir(out), key=lambda f: int(f.split(".")[0]))
    assert pngs == ["%d.png" % i for i in range(16)]
    img = np.asarray(Image.open(os.path.join(out, "7.png")))
    assert img.shape == (64, 64, 3) and img.dtype == np.uint8
