"""Slice / context preparation for transformer training (reference: vidgen/data/dataset_mapper.py:113-149).

Only the `prepare_slices` branch of the reference's mapper is on the hot path (image decoding, dataset
catalogs and samplers are I/O and out of scope).  Given one clip of latent codes (T, nc, H, W) it picks a
random subscale slice (a, b, c) and emits the four tensors VideoTransformerModel consumes.
`prepare_slices_batch` is the batched counterpart that runs on whatever device holds the codes (GPU),
so that the input pipeline does not become host-bound at 8 GPUs.
"""
import random

import numpy as np
import torch

from ..modeling.autoregressive.vt_utils import slice_and_context, subscale_order


def prepare_slices(video_tchw, abc, stride, kernel, n_prime, pad_value=-1):
    """One clip (T, nc, H, W) and a fixed slice offset -> dict(context, slice, slice_idx, ignore_mask)."""
    st, sh, sw = stride
    video = torch.as_tensor(video_tchw)[None].transpose(1, 2)            # 1, nc, T, H, W
    _, nc, T, H, W = video.shape
    assert T % st == 0 and H % sh == 0 and W % sw == 0
    a, b, c = abc
    sl, ctx = slice_and_context(video, a, b, c, stride, kernel, pad_value)
    ignore = torch.zeros(1, 1, T, H, W, dtype=torch.bool)
    if n_prime > 0:
        ignore[:, :, :n_prime] = True
    ignore = ignore[:, :, a::st, b::sh, c::sw].clone()
    _, abc2idx = subscale_order(st, sh, sw)
    return {"context": ctx[0].long(), "slice": sl[0].long(), "slice_idx": torch.tensor(abc2idx[(a, b, c)]).long(),
            "ignore_mask": ignore[0]}


def draw_abc(stride, t_slice, n_prime, rng=random):
    """The reference's random slice choice (dataset_mapper.py:123-127): when a slice is one whole frame
    the first N_PRIME frames are never drawn."""
    st, sh, sw = stride
    single_frame = (t_slice == 1 and sh == 1 and sw == 1)
    a = rng.randint(n_prime, st - 1) if single_frame else rng.randint(0, st - 1)
    return a, rng.randint(0, sh - 1), rng.randint(0, sw - 1)


def prepare_slices_batch(videos_btchw, abcs, stride, kernel, n_prime, pad_value=-1):
    """Batched, device-side builder: videos (B, T, nc, H, W) int64 on any device, abcs list of (a,b,c).
    Returns stacked (context, slice, slice_idx, ignore_mask) ready for compute_supervised_loss.
    On a GPU this is ONE launch of `lvt_slice_context` (a pure index gather); the torch-indexing form below serves
    host tensors (data-loader workers, as in the reference)."""
    if videos_btchw.is_cuda:
        return _prepare_slices_batch_hip(videos_btchw, abcs, stride, kernel, n_prime, pad_value)
    st, sh, sw = stride
    video = videos_btchw.transpose(1, 2)
    B, nc, T, H, W = video.shape
    _, abc2idx = subscale_order(st, sh, sw)
    ctxs, sls, igs = [], [], []
    groups = {}
    for i, abc in enumerate(abcs):
        groups.setdefault(tuple(abc), []).append(i)
    order = []
    for abc, idxs in groups.items():            # one strided gather per distinct slice offset
        sel = torch.as_tensor(idxs, device=video.device)
        sl, ctx = slice_and_context(video.index_select(0, sel), *abc, stride, kernel, pad_value)
        ig = torch.zeros(len(idxs), 1, T, H, W, dtype=torch.bool, device=video.device)
        if n_prime > 0:
            ig[:, :, :n_prime] = True
        ctxs.append(ctx)
        sls.append(sl)
        igs.append(ig[:, :, abc[0]::st, abc[1]::sh, abc[2]::sw])
        order += idxs
    inv = torch.empty(B, dtype=torch.long, device=video.device)
    inv[torch.as_tensor(order, device=video.device)] = torch.arange(B, device=video.device)
    cat = lambda xs: torch.cat(xs, 0).index_select(0, inv).contiguous()     # noqa: E731
    sidx = torch.tensor([abc2idx[tuple(x)] for x in abcs], dtype=torch.long, device=video.device)
    return cat(ctxs).long(), cat(sls).long(), sidx, cat(igs)


def _prepare_slices_batch_hip(videos, abcs, stride, kernel, n_prime, pad_value):
    from ..hip import binding as L
    videos = videos.long().contiguous()
    B, T, nc, H, W = videos.shape
    (st, sh, sw), (kt, kh, kw) = stride, kernel
    assert T % st == 0 and H % sh == 0 and W % sw == 0
    dev = videos.device
    abc = torch.as_tensor([list(x) for x in abcs], dtype=torch.int32).to(dev, non_blocking=True)
    t, h, w = T // st, H // sh, W // sw
    tc, hc, wc = 2 * (kt // 2) + (t - 1) * st + 1, 2 * (kh // 2) + (h - 1) * sh + 1, 2 * (kw // 2) + (w - 1) * sw + 1
    ctx = torch.empty(B, nc, tc, hc, wc, dtype=torch.int64, device=dev)
    sl = torch.empty(B, nc, t, h, w, dtype=torch.int64, device=dev)
    sidx = torch.empty(B, dtype=torch.int64, device=dev)
    ign = torch.empty(B, 1, t, h, w, dtype=torch.bool, device=dev)
    L.check(L.lib().lvt_slice_context(L.ptr(videos), B, T, nc, H, W, L.ptr(abc), st, sh, sw, kt, kh, kw, n_prime,
                                      pad_value, L.ptr(ctx), L.ptr(sl), L.ptr(sidx), L.ptr(ign), L.stream_ptr()),
            "lvt_slice_context")
    return ctx, sl, sidx, ign


class DatasetMapper:
    """Callable with the reference's mapper contract for latent-code clips: takes a dataset dict holding
    `image_sequence` (T, nc, H, W) integer codes and returns the model-ready dict."""

    def __init__(self, cfg, is_train=True):
        self.is_train = is_train
        self.prepare_slices = is_train and cfg.INPUT.PREPARE_SLICES_TRAIN
        v = cfg.MODEL.AUTOREGRESSIVE.VT
        self.stride, self.kernel, self.n_prime, self.pad_value = v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE
        self.n_frames = cfg.INPUT.N_FRAMES_PER_VIDEO_TRAIN if is_train else cfg.INPUT.N_FRAMES_PER_VIDEO_TEST
        self.scale_zeroone = cfg.INPUT.SCALE_TO_ZEROONE

    def _window(self, n):
        if self.n_frames != -1 and n < self.n_frames:
            return None
        start = 0 if (self.n_frames == -1 or not self.is_train) else random.randint(0, n - self.n_frames)
        return slice(start, n if self.n_frames == -1 else start + self.n_frames)

    def __call__(self, dataset_dict):
        d = dict(dataset_dict)
        if "latent_paths" in d or "latent_names" in d:
            # loader records of data/latents.py:get_latent_video_paths (reference dataset_mapper.py:68-77): only the
            # frames of the drawn window are read from disk
            files = d["latent_paths"] if "latent_paths" in d else [d["video_root"] + "/" + n for n in d["latent_names"]]
            win = self._window(len(files))
            if win is None:
                return None
            seq = np.stack([np.load(f) for f in files[win]], axis=0)
        elif "image_sequence" in d:
            seq = np.asarray(d["image_sequence"])
            win = self._window(len(seq))
            if win is None:
                return None                               # too short: the loader retries another index
            seq = seq[win]
        else:
            raise NotImplementedError("only latent-code clips are handled by this mapper (image I/O is out of scope)")
        if not self.prepare_slices:
            d["image_sequence"] = seq
            return d
        assert not self.scale_zeroone
        T = seq.shape[0]
        abc = draw_abc(self.stride, T // self.stride[0], self.n_prime)
        d.update(prepare_slices(seq, abc, self.stride, self.kernel, self.n_prime, self.pad_value))
        d.pop("image_sequence", None)
        return d
