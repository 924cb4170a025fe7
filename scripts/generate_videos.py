#!/usr/bin/env python
"""Sample a 16-frame video from priming frames, with the reference's command line
(scripts/generate_videos.py:103-115):

    python scripts/generate_videos.py --video-dir example --config-file configs/vt/DSFVT.yaml [KEY VAL ...]

VQ-VAE encode of the first N_PRIME frames -> DSFVT autoregressive sampling of the remaining frames -> VQ-VAE
decode -> PNG files `<OUTPUT_DIR>/<frame>.png`.  Checkpoints are read from the paths named in the config
(TEST.VT_SAMPLER.VQ_VAE.*, MODEL.GENERATOR.WEIGHTS); missing files leave the random initialisation in place.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from lvt_amd.config import get_cfg  # noqa: E402
from lvt_amd.modeling import build_model  # noqa: E402
from lvt_amd.utils.checkpoint import Checkpointer  # noqa: E402


def load_video(video_dir, scale_to_zeroone=True):
    """Frames `<idx>.png|jpg` sorted by index -> (T, 3, H, W) float32."""
    files = sorted((f for f in os.listdir(video_dir) if f.split(".")[-1].lower() in ("png", "jpg", "jpeg")),
                   key=lambda f: int(os.path.splitext(f)[0]))
    frames = [np.asarray(Image.open(os.path.join(video_dir, f)).convert("RGB")).transpose(2, 0, 1) for f in files]
    video = np.stack(frames).astype("float32")
    return video / 255.0 if scale_to_zeroone else video


def save_video(video_thwc_u8, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    for i, frame in enumerate(video_thwc_u8):
        Image.fromarray(frame).save(os.path.join(out_dir, "%d.png" % i))


ALLOW_RANDOM_WEIGHTS = False


def _load(module, path):
    """A configured weight file that does not exist is an error (as with the reference's Checkpointer), unless the
    caller asked for random weights explicitly (`--random-weights`, smoke runs without pretrained files)."""
    if path and (os.path.isfile(path) or not ALLOW_RANDOM_WEIGHTS):
        Checkpointer(module).resume_or_load(path, resume=False)          # raises FileNotFoundError when absent
    elif path:
        print("checkpoint %s not found: --random-weights given, keeping the initialised weights" % path)


@torch.no_grad()
def sample_videos(args):
    cfg = get_cfg()
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.TEST.EVALUATORS = "VTSampler"
    cfg.TEST.VT_SAMPLER.NUM_SAMPLES = 1
    vt = build_model(cfg)
    _load(vt.model, cfg.MODEL.GENERATOR.WEIGHTS)
    vt.eval()

    vq_cfg = get_cfg()
    vq_cfg.merge_from_file(cfg.TEST.VT_SAMPLER.VQ_VAE.CFG)
    vq_cfg.MODEL.DEVICE = cfg.MODEL.DEVICE
    vqvae = build_model(vq_cfg)
    _load(vqvae.encoder, cfg.TEST.VT_SAMPLER.VQ_VAE.ENCODER_WEIGHTS)
    _load(vqvae.generator, cfg.TEST.VT_SAMPLER.VQ_VAE.GENERATOR_WEIGHTS)
    _load(vqvae.codebook, cfg.TEST.VT_SAMPLER.VQ_VAE.CODEBOOK_WEIGHTS)
    vqvae.eval()

    n_prime = cfg.TEST.VT_SAMPLER.N_PRIME
    images = load_video(args.video_dir, vq_cfg.INPUT.SCALE_TO_ZEROONE)[:n_prime]
    assert images.shape[0] == n_prime, "need %d priming frames" % n_prime
    print("Loaded %d priming frames" % n_prime)
    latent = vqvae([{"image_sequence": images}])[0]["latent"]              # (n_prime, nc, h, w)
    print("Transferred to latent codes.")
    _, nc, h, w = latent.shape
    seq = latent.new_zeros(16, nc, h, w)
    seq[:n_prime] = latent
    sample = vt([{"image_sequence": seq}])[0]["samples"][0]                # (nc, T, h, w)
    print("Sampled new video.")
    frames = vqvae.decode(sample.transpose(0, 1).contiguous())               # (T, 3, H, W)
    frames = vqvae.back_normalizer(frames)
    if vq_cfg.INPUT.SCALE_TO_ZEROONE:
        frames = frames * 255
    frames = frames.clamp_(0.0, 255.0).permute(0, 2, 3, 1).contiguous().cpu().numpy().astype(np.uint8)
    save_video(frames, cfg.OUTPUT_DIR)
    print("Saved new video to %s" % cfg.OUTPUT_DIR)
    return frames


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Sample video with 16 frames given priming frames")
    parser.add_argument("--config-file", required=True, metavar="FILE")
    parser.add_argument("--video-dir", required=True)
    parser.add_argument("--random-weights", action="store_true",
                        help="run with the initialised weights when a configured checkpoint file is absent")
    parser.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    args = parser.parse_args()
    print("Command Line Args:", args)
    ALLOW_RANDOM_WEIGHTS = args.random_weights
    sample_videos(args)
