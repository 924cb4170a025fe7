"""On-disk latent-code format of the reference (evaluation/codes_extractor.py:36-53, data/datasets/latents.py:10-40):
one int64 `.npy` of shape (nc, h, w) per frame under `<root>/[<class>/]video_<idx>/<frame>.npy`, indexed by a
`latent_video_paths.npy` cache (a pickled list of records) at the dataset root."""
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


CACHE_NAME = "latent_video_paths.npy"


def _natural_key(name):
    # "10.npy" after "9.npy": digit runs compare as integers (the reference's natural_sorted, utils/strings.py:9-23)
    return [int(tok) if tok.isdigit() else tok for tok in re.split(r"(\d+)", name)]


def get_latent_video_paths(root, use_cache=True):
    """Loader records of the reference (data/datasets/latents.py:10-40):
    `{"video_path": dir, "latent_paths": [frame files in natural order], "video_idx": running index}` for every LEAF
    directory under `root` that holds nothing but `.npy` files, visited in `os.walk` order.  With `use_cache` the list
    is read from `<root>/latent_video_paths.npy` when that file exists (whatever the directory holds by now) and
    written there after the first scan -- the same pickle-in-npy container, so caches written by either
    implementation are interchangeable."""
    if not (os.path.isdir(root) or os.path.islink(root)):
        raise AssertionError("%s is not a valid directory" % root)
    cache = os.path.join(root, CACHE_NAME)
    if use_cache and os.path.exists(cache):
        return np.load(cache, allow_pickle=True).tolist()
    records = []
    for cur, dirs, files in os.walk(root):
        if dirs:
            continue                                   # only leaf directories can be videos
        files = sorted(files, key=_natural_key)
        if all(f.endswith(".npy") for f in files):     # (an empty leaf directory is a video of zero frames there too)
            records.append({"video_path": cur, "latent_paths": [os.path.join(cur, f) for f in files],
                            "video_idx": len(records)})
    if use_cache and not os.path.exists(cache):
        np.save(cache, records)
    return records
