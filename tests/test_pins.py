"""Pins for two rows the round-1 review found unpinned (host tests, no GPU):

* A21 weight initialisation (reference ae.py:41-61, vt.py:34-54, vt_attention.py:108-112): a same-seed `build_model`
  must reproduce the reference's tensors -- fixture G19 holds per-tensor checksums captured from the real reference.
* f2 loader records + `latent_video_paths.npy` cache (reference data/datasets/latents.py:10-40) and the mapper fed
  with such a record (dataset_mapper.py:68-77, 113-149) -- fixture G20.
"""
import os
import random

import numpy as np
import pytest
import torch

from conftest import ROOT

LATENT_TREE = {                      # same tree as tests/golden/make_golden.py:LATENT_TREE
    "video_0": ["%d.npy" % i for i in range(12)],
    "video_1": ["0.npy", "1.npy", "2.npy", "10.npy", "9.npy"],
    "clsA/video_7": ["3.npy", "1.npy", "2.npy"],
    "clsA/video_8": ["0.npy"],
    "mixed": ["0.npy", "notes.txt"],
    "emptyleaf": [],
}


def _cfg(path):
    from lvt_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, path))
    cfg.MODEL.DEVICE = "cpu"
    return cfg


def _pins(state_dict):
    names, rows = [], []
    for k, v in state_dict.items():
        if not v.dtype.is_floating_point:
            continue
        d = v.detach().double().reshape(-1)
        names.append(k)
        rows.append([float(d.sum()), float(d.abs().sum()), float(d[0]), float(d[-1]), float(d.numel())])
    return names, np.array(rows, dtype=np.float64)


@pytest.mark.parametrize("tag,path", [("prdvqvae2", "configs/vqvae/PR-DVQVAE2.yaml"),
                                      ("kdvqvae", "configs/vqvae/K-DVQVAE.yaml"),
                                      ("dsfvt", "configs/vt/DSFVT.yaml"), ("dssvt", "configs/vt/DSSVT.yaml")])
def test_init_weights_same_seed_equals_reference(golden, tag, path):
    from lvt_amd.modeling import build_model
    g = golden("g19_init_pins")
    seed = int(g[tag + ".seed"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    model = build_model(_cfg(path))
    parts = {"encoder": model.encoder, "generator": model.generator, "codebook": model.codebook} \
        if hasattr(model, "codebook") else {"model": model.model}
    for part, mod in parts.items():
        names, rows = _pins(mod.state_dict())
        want_names = [str(x) for x in g["%s.%s.names" % (tag, part)]]
        assert names == want_names, (tag, part)                     # same keys, same registration order
        want = g["%s.%s.pins" % (tag, part)].numpy()
        bad = [n for n, a, b in zip(names, rows, want) if not np.array_equal(a, b)]
        assert not bad, (tag, part, bad[:5])                        # bit-identical tensors (checksums in float64)


def _make_tree(root):
    rng = np.random.RandomState(3)
    for leaf, files in LATENT_TREE.items():
        os.makedirs(os.path.join(root, leaf), exist_ok=True)
        for f in files:
            full = os.path.join(root, leaf, f)
            if f.endswith(".npy"):
                np.save(full, rng.randint(0, 512, (4, 16, 16)).astype(np.int64))
            else:
                open(full, "w").write("x")


def test_latent_video_paths_records_and_cache(golden, tmp_path):
    from lvt_amd.data.latents import CACHE_NAME, get_latent_video_paths
    g = golden("g20_latent_paths")
    root = str(tmp_path)
    _make_tree(root)
    recs = get_latent_video_paths(root, use_cache=False)
    assert not os.path.exists(os.path.join(root, CACHE_NAME))
    rel = lambda p: os.path.relpath(p, root)                         # noqa: E731
    assert sorted(recs[0]) == [str(k) for k in g["record_keys"]]
    assert [r["video_idx"] for r in recs] == list(range(len(recs)))  # running index in walk order
    mine = {rel(r["video_path"]): "|".join(rel(q) for q in r["latent_paths"]) for r in recs}
    theirs = dict(zip((str(x) for x in g["video_path"]), (str(x) for x in g["latent_paths"])))
    assert mine == theirs                     # same videos ("mixed" rejected, the empty leaf kept), natural frame order
    assert mine["video_1"].split("|")[-2:] == ["video_1/9.npy", "video_1/10.npy"]
    # cache: written by the first cached scan, then authoritative even when the directory changes
    a = get_latent_video_paths(root, use_cache=True)
    assert a == recs and os.path.exists(os.path.join(root, CACHE_NAME))
    raw = np.load(os.path.join(root, CACHE_NAME), allow_pickle=True).tolist()     # the reference's container
    assert raw == recs
    os.makedirs(os.path.join(root, "video_99"))
    np.save(os.path.join(root, "video_99", "0.npy"), np.zeros((4, 16, 16), np.int64))
    assert get_latent_video_paths(root, use_cache=True) == recs
    assert len(get_latent_video_paths(root, use_cache=False)) == len(recs) + 1
    with pytest.raises(AssertionError):
        get_latent_video_paths(os.path.join(root, "nope"))


def test_mapper_on_loader_record_equals_reference(golden, tmp_path):
    from lvt_amd.data.dataset_mapper import DatasetMapper
    g = golden("g20_latent_paths")
    root = str(tmp_path)
    d0 = os.path.join(root, "video_0")
    os.makedirs(d0)
    for t in range(12):
        np.save(os.path.join(d0, "%d.npy" % t), g["video_0"][t].numpy())
    cfg = _cfg("configs/vt/DSFVT.yaml")
    cfg.INPUT.N_FRAMES_PER_VIDEO_TRAIN = 8
    cfg.MODEL.AUTOREGRESSIVE.VT.STRIDE = (8, 1, 1)
    cfg.MODEL.AUTOREGRESSIVE.VT.N_PRIME = 2
    mapper = DatasetMapper(cfg, True)
    rec = {"video_path": d0, "latent_paths": [os.path.join(d0, "%d.npy" % t) for t in range(12)], "video_idx": 0}
    random.seed(77)
    out = mapper(rec)
    assert sorted(out) == [str(k) for k in g["mapped_keys"]]
    assert torch.equal(out["context"], g["mapped_context"]) and torch.equal(out["slice"], g["mapped_slice"])
    assert torch.equal(out["slice_idx"], g["mapped_slice_idx"]) and torch.equal(out["ignore_mask"], g["mapped_ignore"])
    # the Kinetics flavour of the record (video_root + latent_names) reads the same frames
    random.seed(77)
    out2 = mapper({"video_root": d0, "latent_names": ["%d.npy" % t for t in range(12)], "video_idx": 0, "class": 3})
    assert torch.equal(out2["context"], out["context"]) and torch.equal(out2["slice"], out["slice"])
