"""On-disk latent-code format of the reference (evaluation/codes_extractor.py:36-53, data/datasets/latents.py:10-40):
one int64 `.npy` of shape (nc, h, w) per frame under `<root>/[<class>/]video_<idx>/<frame>.npy`."""
import os
import re

import numpy as np


def save_video_codes(root, video_idx, codes_tchw, class_name=None):
    """codes (T, nc, h, w) integer array -> one file per frame; returns the video directory."""
    d = os.path.join(root, class_name, "video_%d" % video_idx) if class_name else os.path.join(root, "video_%d" % video_idx)
    os.makedirs(d, exist_ok=True)
    codes = np.asarray(codes_tchw).astype(np.int64)
    for t in range(codes.shape[0]):
        np.save(os.path.join(d, "%d.npy" % t), codes[t])
    return d


def list_latent_videos(root):
    """-> sorted list of (video_dir, [frame files in temporal order])."""
    out = []
    for cur, dirs, files in os.walk(root):
        if os.path.basename(cur).startswith("video_"):
            frames = sorted((f for f in files if f.endswith(".npy")), key=lambda f: int(re.sub(r"\D", "", f) or 0))
            if frames:
                out.append((cur, frames))
    return sorted(out)


def load_video_codes(video_dir, frames=None, n_frames=-1):
    if frames is None:
        frames = sorted((f for f in os.listdir(video_dir) if f.endswith(".npy")),
                        key=lambda f: int(re.sub(r"\D", "", f) or 0))
    if n_frames > 0:
        frames = frames[:n_frames]
    return np.stack([np.load(os.path.join(video_dir, f)) for f in frames], axis=0)
