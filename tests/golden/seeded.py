"""Deterministic, device- and library-independent parameter generator for parity tests.

Golden fixtures do NOT store network weights (7-200 MB); they store (seed, inputs, reference
outputs).  The generating script, the oracle tests and the GPU parity tests all rebuild the weights
from `numpy.random.default_rng([seed, crc32(name)])`, which is bit-reproducible everywhere.
Scales follow the reference's initialisers in spirit (xavier-uniform for conv/linear weights,
N(0,1) embeddings) but biases / norm affine / relative-position banks are made non-trivial so that
every term of every kernel is exercised.
"""
import zlib

import numpy as np
import torch

VQVAE_ENCODER_SHAPES = {
    "layers.0.weight": (128, 3, 4, 4), "layers.0.bias": (128,),
    "layers.2.weight": (256, 128, 4, 4), "layers.2.bias": (256,),
    "layers.4.weight": (256, 256, 3, 3), "layers.4.bias": (256,),
    "layers.5.block.1.weight": (128, 256, 3, 3), "layers.5.block.1.bias": (128,),
    "layers.5.block.3.weight": (256, 128, 1, 1), "layers.5.block.3.bias": (256,),
    "layers.6.block.1.weight": (128, 256, 3, 3), "layers.6.block.1.bias": (128,),
    "layers.6.block.3.weight": (256, 128, 1, 1), "layers.6.block.3.bias": (256,),
}
VQVAE_DECODER_SHAPES = {
    "layers.0.weight": (256, 256, 3, 3), "layers.0.bias": (256,),
    "layers.1.block.1.weight": (128, 256, 3, 3), "layers.1.block.1.bias": (128,),
    "layers.1.block.3.weight": (256, 128, 1, 1), "layers.1.block.3.bias": (256,),
    "layers.2.block.1.weight": (128, 256, 3, 3), "layers.2.block.1.bias": (128,),
    "layers.2.block.3.weight": (256, 128, 1, 1), "layers.2.block.3.bias": (256,),
    "layers.4.weight": (256, 128, 4, 4), "layers.4.bias": (128,),   # ConvTranspose2d (in,out,k,k)
    "layers.6.weight": (128, 3, 4, 4), "layers.6.bias": (3,),       # ConvTranspose2d
}


def vqvae_shapes(n_layers=2):
    """(encoder, decoder) state_dict shapes of ResEncoder / ResDecoder (stride 4) with n_layers residual blocks
    (2: PR-DVQVAE2, 4: K-DVQVAE)."""
    enc = {"layers.0.weight": (128, 3, 4, 4), "layers.0.bias": (128,),
           "layers.2.weight": (256, 128, 4, 4), "layers.2.bias": (256,),
           "layers.4.weight": (256, 256, 3, 3), "layers.4.bias": (256,)}
    dec = {"layers.0.weight": (256, 256, 3, 3), "layers.0.bias": (256,)}
    for i in range(n_layers):
        for pre, d in (("layers.%d." % (5 + i), enc), ("layers.%d." % (1 + i), dec)):
            d[pre + "block.1.weight"] = (128, 256, 3, 3)
            d[pre + "block.1.bias"] = (128,)
            d[pre + "block.3.weight"] = (256, 128, 1, 1)
            d[pre + "block.3.bias"] = (256,)
    dec["layers.%d.weight" % (n_layers + 2)] = (256, 128, 4, 4)      # ConvTranspose2d (in,out,k,k)
    dec["layers.%d.bias" % (n_layers + 2)] = (128,)
    dec["layers.%d.weight" % (n_layers + 4)] = (128, 3, 4, 4)
    dec["layers.%d.bias" % (n_layers + 4)] = (3,)
    return enc, dec


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def seeded_array(name, shape, seed):
    r = _rng(seed, name)
    leaf = name.split(".")[-1]
    if "embedding" in name or "ch_embedder" in name:          # nn.Embedding: N(0,1)
        a = r.standard_normal(shape)
    elif leaf.endswith("_bank"):                               # relative-position bias banks
        a = 0.5 * r.standard_normal(shape)
    elif "layer_norm" in name or ".ffn.0." in name:            # LayerNorm affine
        a = (1.0 + 0.1 * r.standard_normal(shape)) if leaf == "weight" else 0.1 * r.standard_normal(shape)
    elif leaf == "bias":
        a = r.uniform(-0.05, 0.05, shape)
    elif leaf in ("w_q", "w_k", "w_v"):                        # (na, d, da): xavier-normal-like
        a = r.standard_normal(shape) * np.sqrt(2.0 / (shape[1] + shape[2]))
    else:                                                      # conv / linear weight: xavier-uniform
        rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        bound = np.sqrt(6.0 / (shape[0] * rf + shape[1] * rf))
        a = r.uniform(-bound, bound, shape)
    return a.astype(np.float32)


def seeded_params(shapes, seed, prefix=""):
    return {k: torch.from_numpy(seeded_array(prefix + k, s, seed)) for k, s in shapes.items()}


def seeded_codebook_state(seed, num=4, K=512, D=64, scale=1.0):
    """Codebook state with de-aliased running_sum (GPU semantics) and a non-trivial running_size."""
    st = {}
    for i in range(num):
        r = _rng(seed, "ve.%d" % i)
        w = (scale * r.standard_normal((K, D))).astype(np.float32)
        n = r.uniform(0.5, 4.0, (K,)).astype(np.float32)
        st["ve.%d.embedding.weight" % i] = torch.from_numpy(w)
        st["ve.%d.running_size" % i] = torch.from_numpy(n)
        st["ve.%d.running_sum" % i] = torch.from_numpy(w * n[:, None])
    return st


def seeded_input(name, shape, seed, lo=0.0, hi=1.0):
    return torch.from_numpy(_rng(seed, name).uniform(lo, hi, shape).astype(np.float32))


def seeded_codes(name, shape, seed, nv=512):
    return torch.from_numpy(_rng(seed, name).integers(0, nv, shape, dtype=np.int64))


def dsfvt_shapes(nc=4, nv=512, de=128, d=512, da=128, na=8, n_enc=8, n_dec=8, block=(1, 16, 16),
                 kernel=(7, 1, 1), n_slices=16, class_num=0):
    """state_dict parameter shapes of VideoTransformer (parameters only).  Defaults: DSFVT; DSSVT is
    block (4,8,8), kernel (1,3,3), n_slices 4; DSTSVT is block (4,8,8), kernel (5,3,3), n_slices 16."""
    t, h, w = block
    s = {"encoder.conv.weight": (de, nc * nv) + tuple(kernel), "encoder.conv.bias": (de,),
         "encoder.slice_embedding.weight": (n_slices, de),
         "encoder.linear_projector.weight": (d, de * (2 if class_num > 0 else 1), 1, 1, 1)}
    if class_num > 0:
        s["encoder.class_embedding.weight"] = (class_num, de)
    for i in range(nc):
        s["decoder.ch_embedder.%d.weight" % i] = (nv, de)
    s["decoder.conv.conv.weight"] = (d, de, 3, 3, 3)
    s["decoder.conv.conv.bias"] = (d,)
    s["decoder.linear_projector.weight"] = (d, d, 1, 1, 1)
    for side, n in (("encoder", n_enc), ("decoder", n_dec)):
        for i in range(n):
            p = "%s.block_local_attention.%d." % (side, i)
            s[p + "dt_bank"] = (na, 2 * t - 1)
            s[p + "dh_bank"] = (na, 2 * h - 1)
            s[p + "dw_bank"] = (na, 2 * w - 1)
            for nm in ("w_q", "w_k", "w_v"):
                s[p + "mha." + nm] = (na, d, da)
            s[p + "mha.layer_norm.weight"] = (d,)
            s[p + "mha.layer_norm.bias"] = (d,)
            s[p + "mha.proj.weight"] = (d, na * da)
            s[p + "ffn.0.weight"] = (d,)
            s[p + "ffn.0.bias"] = (d,)
            s[p + "ffn.1.weight"] = (d, d)
            s[p + "ffn.1.bias"] = (d,)
            s[p + "ffn.3.weight"] = (d, d)
            s[p + "ffn.3.bias"] = (d,)
    s["ch_predictor.layer_norm.weight"] = (d,)
    s["ch_predictor.layer_norm.bias"] = (d,)
    for k in range(nc):
        s["ch_predictor.U.%d.weight" % k] = (d, d + k * nv)
        s["ch_predictor.U.%d.bias" % k] = (d,)
        s["ch_predictor.P.%d.weight" % k] = (nv, d)
        s["ch_predictor.P.%d.bias" % k] = (nv,)
    return s
