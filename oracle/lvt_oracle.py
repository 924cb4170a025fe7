"""CPU oracle for the LVT hot path (VQ-VAE + DSFVT latent transformer).

TEST INFRASTRUCTURE ONLY.  Nothing under `lvt_amd/` may import this module; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and only as the checker /
the timed CPU baseline -- never as (part of) the product path.

It is a *functional* restatement, in plain PyTorch-CPU fp32, of the algorithms the reference
(`rakhimovv/lvt`, package `vidgen`, mounted at /root/reference in the build container) executes on
the path named by BASELINE.json.  Every function cites the reference file:line it follows.  All
parameters are passed explicitly as dicts whose keys are the reference's `state_dict()` keys, so
the same seeded tensors can be fed to the reference (golden capture), to this oracle and to the
HIP product path.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real reference (through
`oracle/shim`) in the build container, runs it on seeded inputs and commits input/output vectors
under `tests/golden/`; `tests/test_oracle_golden.py` checks this oracle against those vectors
(bit-exact for indices, ~1e-6 for floats) on every CPU test run.

Third-party arithmetic: everything is PyTorch CPU (ATen -> oneDNN / MKL); the reference does not
pin a torch version (setup.py:12-13).  Goldens record torch.__version__ and the thread count.
"""
import math

import torch
import torch.nn.functional as F

__all__ = [
    "normalize", "back_normalize", "res_block", "res_encoder", "res_decoder", "vq_nearest",
    "vq_straight_through", "vq_ema_step", "dvq_indices", "dvq_straight_through", "dvq_embed",
    "vqvae_supervised_loss", "vqvae_encode", "vqvae_decode", "vqvae_inference",
    "subscale_order", "slice_mask", "visible_abc_mask", "ss_shift", "prepare_slices",
    "positional_encoding_table", "rel_position_bias", "layer_norm", "multi_head_attention",
    "block_local_attention", "masked_conv3d", "vt_encoder", "vt_decoder",
    "channel_predictor_logits", "channel_predictor_pixel_probs", "video_transformer_logits",
    "vt_supervised_loss", "vt_logits_for_entire_video", "multinomial_from_uniform",
]


# --------------------------------------------------------------------------------------------
# A1  input normalisation                      (meta_arch/ae.py:32-37, :151-168)
# --------------------------------------------------------------------------------------------
def normalize(x, mean, std):
    """(x - mean) / std per channel; x is (N,C,H,W).  ae.py:36."""
    c = len(mean)
    m = torch.tensor(mean, dtype=torch.float32).view(1, c, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, c, 1, 1)
    return (x - m) / s


def back_normalize(y, mean, std):
    """y * std + mean.  ae.py:37."""
    c = len(mean)
    m = torch.tensor(mean, dtype=torch.float32).view(1, c, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, c, 1, 1)
    return y * s + m


# --------------------------------------------------------------------------------------------
# A2 / A7  conv encoder / decoder              (encoder/resencoder.py:10-76, generator/resdecoder.py:10-75)
# --------------------------------------------------------------------------------------------
def res_block(p, prefix, x):
    """ResBlock with the reference's in-place first ReLU (resencoder.py:13-21).

    `nn.ReLU(True)` is the first op of `block` and mutates the block input, so the value added by
    the skip connection is relu(x), not x:   out = relu(x) + conv1x1(relu(conv3x3(relu(x)))).
    """
    a = torch.relu(x)
    t = F.conv2d(a, p[prefix + "block.1.weight"], p[prefix + "block.1.bias"], stride=1, padding=1)
    t = torch.relu(t)
    t = F.conv2d(t, p[prefix + "block.3.weight"], p[prefix + "block.3.bias"])
    return a + t


def res_encoder(p, x, n_layers=2):
    """ResEncoder, stride 4, NORM "", no out activation (resencoder.py:45-52, 60-76)."""
    h = F.conv2d(x, p["layers.0.weight"], p["layers.0.bias"], stride=2, padding=1)
    h = torch.relu(h)
    h = F.conv2d(h, p["layers.2.weight"], p["layers.2.bias"], stride=2, padding=1)
    h = torch.relu(h)
    h = F.conv2d(h, p["layers.4.weight"], p["layers.4.bias"], stride=1, padding=1)
    for i in range(n_layers):
        h = res_block(p, "layers.%d." % (5 + i), h)
    return h


def res_decoder(p, z, n_layers=2, out_activation="tanh"):
    """ResDecoder, stride 4 (resdecoder.py:48-75): conv3x3, ResBlocks, ReLU, ConvT, ReLU, ConvT, tanh."""
    h = F.conv2d(z, p["layers.0.weight"], p["layers.0.bias"], stride=1, padding=1)
    for i in range(n_layers):
        h = res_block(p, "layers.%d." % (1 + i), h)
    h = torch.relu(h)
    k = 1 + n_layers + 1
    h = F.conv_transpose2d(h, p["layers.%d.weight" % k], p["layers.%d.bias" % k], stride=2, padding=1)
    h = torch.relu(h)
    k += 2
    h = F.conv_transpose2d(h, p["layers.%d.weight" % k], p["layers.%d.bias" % k], stride=2, padding=1)
    if out_activation == "tanh":
        h = torch.tanh(h)
    elif out_activation == "sigmoid":
        h = torch.sigmoid(h)
    elif out_activation != "":
        raise ValueError(out_activation)
    return h


# --------------------------------------------------------------------------------------------
# A3 / A4  nearest-codebook lookup             (vq/vq_utils.py:5-25, :34-65)
# --------------------------------------------------------------------------------------------
def vq_nearest(x_last, codebook):
    """argmin_k ( |e_k|^2 + |x|^2 - 2 x.e_k ), x_last (..., D) channels-last, codebook (K, D).

    Follows vq_utils.py:13-20 op for op (sum of squares, addmm(beta=1, alpha=-2), torch.min(dim=1))
    so that ties / near-ties resolve exactly as in the reference on the same CPU build.
    Returns int64 indices of shape x_last.shape[:-1].
    """
    d = codebook.size(1)
    flat = x_last.reshape(-1, d)
    cb_sqr = torch.sum(codebook ** 2, dim=1)
    in_sqr = torch.sum(flat ** 2, dim=1, keepdim=True)
    dist = torch.addmm(cb_sqr + in_sqr, flat, codebook.t(), alpha=-2.0, beta=1.0)
    _, idx = torch.min(dist, dim=1)
    return idx.view(*x_last.shape[:-1])


def vq_margin_fp64(x_last, codebook):
    """fp64 best / second-best squared distances per row (test helper for near-tie accounting)."""
    flat = x_last.reshape(-1, codebook.size(1)).double()
    e = codebook.double()
    dist = (e ** 2).sum(1)[None, :] + (flat ** 2).sum(1, keepdim=True) - 2.0 * flat @ e.t()
    top2 = torch.topk(dist, 2, dim=1, largest=False)
    return top2.values[:, 0], top2.values[:, 1], top2.indices[:, 0]


def vq_straight_through(x_last, codebook):
    """vq_st forward (vq_utils.py:36-46): (codes gathered from `codebook`, flat indices)."""
    idx = vq_nearest(x_last, codebook).view(-1)
    codes = torch.index_select(codebook, 0, idx).view_as(x_last)
    return codes, idx


# --------------------------------------------------------------------------------------------
# A5  EMA codebook update                      (vq/vq_embedding.py:35-66)
# --------------------------------------------------------------------------------------------
def vq_ema_step(state, z_e_part, decay=0.99, eps=1e-5, all_reduce=None, alias_running_sum=False,
                force_idx=None, ema=True):
    """One `_straight_through` call of VQEmbedding (vq_embedding.py:35-66).  ema=False (CODEBOOK.EMA False: the codebook
    is a trained parameter): no update, z_q_bar is gathered from the codebook WITH its graph (vq_embedding.py:61-64), the
    straight-through value from the detached one (:37).

    state: dict with 'embedding.weight' (K,D), 'running_size' (K,), 'running_sum' (K,D).
    z_e_part: (N, D, H, W).  Returns (z_q_st, z_q_bar, new_state, idx_flat).
      z_q_st  gathered from the PRE-update codebook  (vq_embedding.py:37-38)
      z_q_bar gathered from the POST-update codebook (vq_embedding.py:61-64)
    all_reduce: optional callable summing a tensor over ranks (vq_embedding.py:46-47, 53-54).
    alias_running_sum: reproduce the reference's CPU-only quirk where `running_sum` shares storage
      with `embedding.weight` (vq_embedding.py:21): the EMA then decays the *current codebook*
      rather than a separate running sum, and the final copy_ overwrites it.  On GPU `.to(device)`
      de-aliases, which is the behaviour the product implements (alias_running_sum=False).
    """
    w = state["embedding.weight"]
    k = w.size(0)
    x = z_e_part.permute(0, 2, 3, 1).contiguous()
    codes, idx = vq_straight_through(x, w)
    if force_idx is not None:
        # test hook: continue with externally chosen indices (isolates tie-breaking from the rest)
        idx = force_idx.reshape(-1)
        codes = torch.index_select(w, 0, idx).view_as(x)
    z_q_st = codes.permute(0, 3, 1, 2).contiguous()
    if not ema:
        z_q_bar = torch.index_select(w, 0, idx).view_as(x).permute(0, 3, 1, 2).contiguous()
        return z_q_st.detach(), z_q_bar, dict(state), idx

    size = torch.zeros(k, dtype=torch.int64)
    size.index_add_(0, idx, torch.ones_like(idx))
    if all_reduce is not None:
        size = all_reduce(size)
    running_size = state["running_size"] * decay + (1 - decay) * size

    total = torch.zeros_like(w)
    total.index_add_(0, idx, x.view(-1, w.size(1)))
    if all_reduce is not None:
        total = all_reduce(total)
    base = w if alias_running_sum else state["running_sum"]
    running_sum = base * decay + (1 - decay) * total

    n = running_size.sum()
    size_ = (running_size + eps) / (n + k * eps) * n
    new_w = running_sum / size_.unsqueeze(1)
    new_state = {
        "embedding.weight": new_w,
        "running_size": running_size,
        "running_sum": new_w.clone() if alias_running_sum else running_sum,
    }
    z_q_bar = torch.index_select(new_w, 0, idx).view_as(x).permute(0, 3, 1, 2).contiguous()
    return z_q_st, z_q_bar, new_state, idx


# --------------------------------------------------------------------------------------------
# A6  product quantiser                        (vq/vq_embedding.py:69-99)
# --------------------------------------------------------------------------------------------
def _split_state(state, i):
    pre = "ve.%d." % i
    return {k[len(pre):]: v for k, v in state.items() if k.startswith(pre)}


def dvq_indices(state, z_e, num=4):
    """mode "": (N,D,H,W) -> (N,num,H,W) int64 (vq_embedding.py:79-83, :25-28)."""
    assert z_e.dim() == 4 and z_e.size(1) % num == 0
    out = []
    for i, part in enumerate(z_e.split(z_e.size(1) // num, dim=1)):
        w = state["ve.%d.embedding.weight" % i]
        out.append(vq_nearest(part.permute(0, 2, 3, 1).contiguous(), w))
    return torch.stack(out, dim=1)


def dvq_straight_through(state, z_e, num=4, all_reduce=None, alias_running_sum=False, force_idx=None, ema=True):
    """mode "st" (vq_embedding.py:84-91).  Returns (z_q_st, z_q_bar, new_state, idx (num, N*H*W))."""
    assert z_e.dim() == 4 and z_e.size(1) % num == 0
    r1, r2, idxs, new_state = [], [], [], {}
    for i, part in enumerate(z_e.split(z_e.size(1) // num, dim=1)):
        a, b, ns, idx = vq_ema_step(_split_state(state, i), part, all_reduce=all_reduce,
                                    alias_running_sum=alias_running_sum,
                                    force_idx=None if force_idx is None else force_idx[:, i], ema=ema)
        r1.append(a)
        r2.append(b)
        idxs.append(idx)
        for k, v in ns.items():
            new_state["ve.%d.%s" % (i, k)] = v
    return torch.cat(r1, 1), torch.cat(r2, 1), new_state, torch.stack(idxs, 0)


def dvq_embed(state, latents, num=4):
    """mode "emb": (N,num,H,W) int64 -> (N,H,W,D) (vq_embedding.py:92-97)."""
    parts = [F.embedding(latents[:, i], state["ve.%d.embedding.weight" % i]) for i in range(num)]
    return torch.cat(parts, dim=-1)


def single_codebook_state(state):
    """CODEBOOK.NUM == 1 (meta_arch/vqvae.py:25-27): the model's quantiser IS one `VQEmbedding` (vq_embedding.py:9-66) whose state
    has no `ve.i.` prefix.  It is the product quantiser with ONE part -- `split(D, dim=1)` of a D-channel tensor is the tensor,
    `cat` / `stack` of one piece is the piece -- so every dvq_* function restates it with num = 1 on the re-keyed state; the
    latents of the single form have no codebook axis (squeeze dim 1 of dvq_indices)."""
    return {"ve.0." + k: v for k, v in state.items()}


# --------------------------------------------------------------------------------------------
# A8  VQ-VAE meta-architecture                 (meta_arch/vqvae.py:66-106, loss/loss.py:5-20)
# --------------------------------------------------------------------------------------------
class _StraightThrough(torch.autograd.Function):
    """Value of the quantised tensor, gradient copied to z_e (vq_utils.py:52-54)."""

    @staticmethod
    def forward(ctx, z_e, z_q):
        return z_q.clone()

    @staticmethod
    def backward(ctx, g):
        return g.clone(), None


def vqvae_supervised_loss(enc, dec, cb_state, x, beta=1.0, lam=1.0, num=4, all_reduce=None,
                          alias_running_sum=False, force_idx=None, n_layers=2, ema=True, pixel_mode="l2"):
    """compute_supervised_loss (vqvae.py:66-91).  x already normalised, (N,3,H,W) or (B,T,3,H,W).
    n_layers = residual blocks per side (2: PR-DVQVAE2, 4: K-DVQVAE).

    Returns (loss_dict, new_codebook_state, aux) with aux = dict(z_e, z_q_st, x_tilde, idx).
    """
    if x.dim() == 5:
        b, t, c, h, w = x.shape
        x = x.reshape(b * t, c, h, w)
    z_e = res_encoder(enc, x, n_layers)
    z_q_st_val, z_q_bar, new_state, idx = dvq_straight_through(
        cb_state, z_e.detach(), num, all_reduce, alias_running_sum, force_idx, ema)
    z_q_st = _StraightThrough.apply(z_e, z_q_st_val)
    x_tilde = res_decoder(dec, z_q_st, n_layers)
    losses = {
        # PixelLoss (loss/loss.py:10-19): LOSS.PIXEL.MODE "l2" -> mse_loss, "l1" -> l1_loss, times LAMBDA
        "loss_reconstruction": lam * (F.l1_loss if pixel_mode == "l1" else F.mse_loss)(x_tilde, x),
        "loss_commitment": beta * F.mse_loss(z_e, z_q_bar.detach()),
    }
    if not ema:
        losses["loss_dict"] = F.mse_loss(z_q_bar, z_e.detach())      # (the reference's key name: vqvae.py:83-84)
    return losses, new_state, {"z_e": z_e, "z_q_st": z_q_st_val, "x_tilde": x_tilde, "idx": idx}


def vqvae_encode(enc, cb_state, x, num=4):
    """encode (vqvae.py:93-101): (N,3,H,W) -> (N,num,h,w) int64."""
    return dvq_indices(cb_state, res_encoder(enc, x), num)


def vqvae_decode(dec, cb_state, latents, num=4):
    """decode (vqvae.py:103-106): (N,num,h,w) int64 -> (N,3,H,W)."""
    z_q = dvq_embed(cb_state, latents, num).permute(0, 3, 1, 2).contiguous()
    return res_decoder(dec, z_q)


def vqvae_inference(enc, dec, cb_state, x01, mean, std, num=4):
    """mode='inference' on a 4-D batch in [0,1] (ae.py:120-147): (reconstruction in [0,1], latent)."""
    x = normalize(x01, mean, std)
    latent = vqvae_encode(enc, cb_state, x, num)
    out = back_normalize(vqvae_decode(dec, cb_state, latent, num), mean, std).clamp_(0.0, 1.0)
    return out, latent


# --------------------------------------------------------------------------------------------
# A19 / A20  subscale helpers and the slice/context builder
#            (autoregressive/vt_utils.py:6-14, 24-33, 48-57, 104-128; data/dataset_mapper.py:113-149)
# --------------------------------------------------------------------------------------------
def subscale_order(st, sh, sw):
    """Raster order over (a,b,c) (vt_utils.py:6-14)."""
    idx2abc = [(a, b, c) for a in range(st) for b in range(sh) for c in range(sw)]
    return idx2abc, {abc: i for i, abc in enumerate(idx2abc)}


def slice_mask(a, b, c, st, sh, sw, T, H, W, dtype=torch.bool):
    """1 at positions == (a,b,c) mod (st,sh,sw) (vt_utils.py:24-33).  Shape (1,1,T,H,W)."""
    m = torch.zeros(1, 1, T, H, W, dtype=dtype)
    m[0, 0, a::st, b::sh, c::sw] = 1
    return m


def visible_abc_mask(a, b, c, st, sh, sw, T, H, W, dtype=torch.bool):
    """Union of the slices strictly before (a,b,c) in subscale order (vt_utils.py:48-57)."""
    idx2abc, abc2idx = subscale_order(st, sh, sw)
    m = torch.zeros(1, 1, T, H, W, dtype=torch.int64)
    for (ai, bi, ci) in idx2abc[:abc2idx[(a, b, c)]]:
        m[0, 0, ai::st, bi::sh, ci::sw] += 1
    return m.to(dtype)


def ss_shift(x, a, b, c, st, sh, sw, T, H, W, kt, kh, kw, pad_value=0):
    """Crop/pad so a stride-(st,sh,sw) conv's first window is centred on slice element (a,b,c)
    (vt_utils.py:104-128)."""
    crops, pads = [], []
    for off, s, n, k in ((a, st, T, kt), (b, sh, H, kh), (c, sw, W, kw)):
        cnt = n // s
        first, last = off, off + (cnt - 1) * s
        lo, hi = k // 2 - first, k // 2 - (n - last - 1)
        crops.append((max(0, -lo), n - max(0, -hi)))
        pads.append((max(0, lo), max(0, hi)))
    x = x[:, :, crops[0][0]:crops[0][1], crops[1][0]:crops[1][1], crops[2][0]:crops[2][1]]
    pad = [pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]]
    return F.pad(x, pad=pad, mode="constant", value=pad_value)


def prepare_slices(video_tchw, abc, stride, kernel, n_prime, pad_value=-1):
    """DatasetMapper's `prepare_slices` branch for a forced (a,b,c) (dataset_mapper.py:113-149).

    video_tchw: (T, nc, H, W) int codes.  Returns dict(context, slice, slice_idx, ignore_mask).
    """
    st, sh, sw = stride
    video = torch.as_tensor(video_tchw)[None].transpose(1, 2)  # 1, nc, T, H, W
    _, nc, T, H, W = video.shape
    t, h, w = T // st, H // sh, W // sw
    a, b, c = abc
    _, abc2idx = subscale_order(st, sh, sw)
    smask = slice_mask(a, b, c, st, sh, sw, T, H, W)
    sl = video.masked_select(smask).clone().view(1, nc, t, h, w)
    vmask = visible_abc_mask(a, b, c, st, sh, sw, T, H, W)
    ctx = ss_shift(video.masked_fill(~vmask, pad_value), a, b, c, st, sh, sw, T, H, W, *kernel,
                   pad_value=pad_value)
    ignore = torch.zeros(1, 1, T, H, W, dtype=torch.bool)
    if n_prime > 0:
        ignore[:, :, :n_prime] = True
    ignore = ignore.masked_select(smask).clone().view(1, 1, t, h, w)
    return {"context": ctx[0].long(), "slice": sl[0].long(),
            "slice_idx": torch.tensor(abc2idx[(a, b, c)]).long(), "ignore_mask": ignore[0]}


# --------------------------------------------------------------------------------------------
# A12  positional encoding                     (autoregressive/vt_attention.py:10-50)
# --------------------------------------------------------------------------------------------
def positional_encoding_table(d_model, T, H, W, min_timescale=1.0, max_timescale=1.0e4):
    """The (d_model, T, H, W) signal that PositionalEncoding.forward adds in place."""
    num_dims = 3
    nts = d_model // (num_dims * 2)
    inc = math.log(max_timescale / min_timescale) / nts  # np.log on python floats == math.log
    inv = min_timescale * torch.exp(torch.arange(nts).float() * -inc)
    table = torch.zeros(d_model, T, H, W)
    for dim, length in enumerate((T, H, W)):
        pos = torch.arange(length, dtype=torch.float)
        scaled = pos.view(-1, 1) * inv.view(1, -1)
        sig = torch.cat([torch.sin(scaled), torch.cos(scaled)], 1)  # (length, 2*nts)
        shape = [1, 1, 1]
        shape[dim] = length
        table[dim * 2 * nts:(dim + 1) * 2 * nts] += sig.t().reshape(2 * nts, *shape)
    return table


# --------------------------------------------------------------------------------------------
# A13 / A14  attention layer                   (autoregressive/vt_attention.py:52-202)
# --------------------------------------------------------------------------------------------
def rel_position_bias(dt_bank, dh_bank, dw_bank, block):
    """get_B (vt_attention.py:169-174): (na, 1, S, S) with S = t*h*w, index = delta + (size-1)."""
    t, h, w = block
    it = torch.arange(t).view(t, 1, 1).expand(t, h, w).reshape(-1)
    ih = torch.arange(h).view(1, h, 1).expand(t, h, w).reshape(-1)
    iw = torch.arange(w).view(1, 1, w).expand(t, h, w).reshape(-1)
    s = t * h * w
    dt = (it[:, None] - it[None, :] + (t - 1)).reshape(-1)
    dh = (ih[:, None] - ih[None, :] + (h - 1)).reshape(-1)
    dw = (iw[:, None] - iw[None, :] + (w - 1)).reshape(-1)
    na = dt_bank.size(0)
    return (dt_bank.index_select(1, dt) + dh_bank.index_select(1, dh)
            + dw_bank.index_select(1, dw)).view(na, 1, s, s)


def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.size(-1),), w, b, eps)


def multi_head_attention(p, pre, x, B, masked):
    """MultiHeadAttention.forward (vt_attention.py:114-129) + ScaledDotProductAttention (:59-81).

    x (b, S, d).  softmax(masked_fill(q k^T / sqrt(da) + B, triu(1), -1e4)) v, heads concatenated
    head-major, Linear(na*da -> d, no bias), + residual.
    """
    b, s, d = x.shape
    w_q, w_k, w_v = p[pre + "w_q"], p[pre + "w_k"], p[pre + "w_v"]
    na, _, da = w_q.shape
    xn = layer_norm(x, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"]).reshape(b * s, d)
    q = torch.matmul(xn, w_q).view(na, b, s, da)
    k = torch.matmul(xn, w_k).view(na, b, s, da)
    v = torch.matmul(xn, w_v).view(na, b, s, da)
    attn = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(da)
    attn = attn + B
    if masked:
        m = torch.triu(torch.ones(1, 1, s, s), diagonal=1).bool()   # reference: float buffer .bool()
        attn = attn.masked_fill(m, -1e4)
    attn = torch.softmax(attn, dim=3)
    out = torch.matmul(attn, v)  # na, b, s, da
    out = out.permute(1, 2, 0, 3).reshape(b, s, na * da)
    return F.linear(out, p[pre + "proj.weight"]) + x


def block_local_attention(p, pre, x, block, masked):
    """BlockLocalAttention.forward (vt_attention.py:176-202).

    x: (b, C, T, H, W).  (T,H,W) == block: the whole slice is one block (:182-188).  Otherwise (:189-200) the
    volume is cut into (T/t)*(H/h)*(W/w) blocks of (t,h,w) tokens that are attended independently (folded into
    the batch dimension, block order (st, sh, sw)) and put back in place.
    ffn = LN -> Linear -> ReLU -> Linear, + x (:138,186).
    """
    b, c, T, H, W = x.shape
    t, h, w = block
    B = rel_position_bias(p[pre + "dt_bank"], p[pre + "dh_bank"], p[pre + "dw_bank"], block)

    def layer(tok):
        tok = multi_head_attention(p, pre + "mha.", tok, B, masked)
        f = layer_norm(tok, p[pre + "ffn.0.weight"], p[pre + "ffn.0.bias"])
        f = torch.relu(F.linear(f, p[pre + "ffn.1.weight"], p[pre + "ffn.1.bias"]))
        f = F.linear(f, p[pre + "ffn.3.weight"], p[pre + "ffn.3.bias"])
        return f + tok

    if (T, H, W) == (t, h, w):
        tok = layer(x.view(b, c, -1).transpose(1, 2).contiguous())
        return tok.transpose(1, 2).contiguous().view(b, c, T, H, W)
    nt, nh, nw = T // t, H // h, W // w
    xb = x.view(b, c, nt, t, nh, h, nw, w).permute(0, 2, 4, 6, 1, 3, 5, 7)      # b, nt, nh, nw, c, t, h, w
    tok = layer(xb.reshape(b * nt * nh * nw, c, t * h * w).transpose(1, 2).contiguous())
    xb = tok.transpose(1, 2).reshape(b, nt, nh, nw, c, t, h, w).permute(0, 4, 1, 5, 2, 6, 3, 7)
    return xb.contiguous().view(b, c, T, H, W)


# --------------------------------------------------------------------------------------------
# A10  masked (causal) 3-D conv                (autoregressive/vt_utils.py:183-200)
# --------------------------------------------------------------------------------------------
def masked_conv3d(weight, bias, x):
    """pad (w: k//2,k//2; h: k-1,0; t: k-1,0), zero taps [:,:,-1,-1,kw//2:], conv3d.

    Returns (y, masked_weight); the reference writes the zeroed taps back into weight.data.
    """
    kt, kh, kw = weight.shape[2:]
    # The reference zeroes the taps in weight.data (no autograd mask), so the gradient w.r.t. the
    # zeroed taps is the plain conv weight-gradient, not 0.  value: zeroed; gradient: identity.
    tap = torch.zeros_like(weight)
    if kw // 2 > 0:
        tap[:, :, -1, -1, kw // 2:] = 1
    wm = weight - (weight * tap).detach()
    xp = F.pad(x, [kw // 2, kw // 2, kh - 1, 0, kt - 1, 0])
    return F.conv3d(xp, wm, bias), wm


# --------------------------------------------------------------------------------------------
# A9 / A11 / A15 / A16  the video transformer  (autoregressive/videotransformer.py)
# --------------------------------------------------------------------------------------------
def vt_encoder(p, context, slice_idx, blocks, stride, nv=512, pad_value=-1, class_idx=None):
    """VTEncoder.forward (videotransformer.py:35-59).  With class_idx (class_num > 0, :54-56) the class
    embedding is broadcast over the volume and concatenated on the channel axis before the 2*de -> d projector.

    context (b, nc, T', H, W) int64 with pads == pad_value.  One-hot (pads -> all-zero rows),
    Conv3d(nc*nv -> de, kernel, stride, bias), + slice_embedding, 1x1x1 projector (no bias),
    unmasked attention stack.  `positional_encoder` is never applied by the reference.
    """
    pre = "encoder."
    mask = context == pad_value
    oh = F.one_hot(context.masked_fill(mask, 0), nv)           # b,nc,T,H,W,nv
    oh = oh.masked_fill(mask.unsqueeze(-1), 0)
    oh = oh.permute(0, 1, 5, 2, 3, 4).contiguous()
    b, nc, _, T, H, W = oh.shape
    w = p[pre + "conv.weight"]
    x = F.conv3d(oh.view(b, nc * nv, T, H, W).to(w.dtype), w, p[pre + "conv.bias"], stride=stride)
    x = x + F.embedding(slice_idx, p[pre + "slice_embedding.weight"])[:, :, None, None, None]
    if class_idx is not None:
        ce = F.embedding(class_idx, p[pre + "class_embedding.weight"])[:, :, None, None, None].expand_as(x)
        x = torch.cat([x, ce], dim=1)
    x = F.conv3d(x, p[pre + "linear_projector.weight"])
    for i, blk in enumerate(blocks):
        x = block_local_attention(p, pre + "block_local_attention.%d." % i, x, blk, masked=False)
    return x


def vt_decoder(p, sl, zl, blocks):
    """VTDecoder.forward (videotransformer.py:80-101)."""
    pre = "decoder."
    b, nc, t, h, w = sl.shape
    emb = 0
    for k in range(nc):
        emb = emb + F.embedding(sl[:, k], p[pre + "ch_embedder.%d.weight" % k])   # b,t,h,w,de
    x = emb.permute(0, 4, 1, 2, 3)
    x, _ = masked_conv3d(p[pre + "conv.conv.weight"], p[pre + "conv.conv.bias"], x)
    x = x + positional_encoding_table(x.size(1), t, h, w)[None].to(x.dtype)
    x = x + F.conv3d(zl, p[pre + "linear_projector.weight"])
    for i, blk in enumerate(blocks):
        x = block_local_attention(p, pre + "block_local_attention.%d." % i, x, blk, masked=True)
    return x


def channel_predictor_logits(p, sl, yl, nv=512, ch_embedder=None):
    """ChannelPredictor 'logits' (videotransformer.py:139-160): per-channel output layers P.k (SHARE_P False) or, when the
    parameter dict holds `ch_predictor.P.weight`, the ONE shared layer of SHARE_P True (:121-123,150-151).
    ch_embedder (list of nc (nv, de) tables): SHARE_EMBEDDINGS (:124-125,152-154) -- the shared P maps d -> de and the
    decoder's channel embedding table is the output matrix, `F.linear(out, weight=ch_embedder[k].weight)`.
    Returns list of nc tensors (b, nv, t, h, w)."""
    pre = "ch_predictor."
    b, d, t, h, w = yl.shape
    nc = sl.size(1)
    y = layer_norm(yl.view(b, d, -1).transpose(1, 2), p[pre + "layer_norm.weight"],
                   p[pre + "layer_norm.bias"])
    oh = F.one_hot(sl.view(b, nc, -1).transpose(1, 2), nv).view(b, t * h * w, nc * nv).to(y.dtype)
    out = []
    for k in range(nc):
        inp = y if k == 0 else torch.cat((y, oh[:, :, :k * nv]), dim=2)
        u = F.linear(inp, p[pre + "U.%d.weight" % k], p[pre + "U.%d.bias" % k])
        pk = "P." if (pre + "P.weight") in p else "P.%d." % k
        o = F.linear(torch.relu(u), p[pre + pk + "weight"], p[pre + pk + "bias"])
        if ch_embedder is not None:
            o = F.linear(o, ch_embedder[k])
        out.append(o.transpose(1, 2).contiguous().view(b, nv, t, h, w))
    return out


def multinomial_from_uniform(prob, u):
    """Inverse-CDF draw used instead of torch.multinomial so sampling is reproducible across
    devices: smallest j with cumsum(prob)[j] > u * sum(prob).  prob (b, nv), u (b,) in [0,1)."""
    cdf = torch.cumsum(prob, dim=1)
    thr = (u * cdf[:, -1]).unsqueeze(1)
    return torch.clamp((cdf <= thr).sum(dim=1), max=prob.size(1) - 1)


def channel_predictor_pixel_probs(p, yl, pixel, uniforms, nv=512, temp=1.0):
    """ChannelPredictor 'sample_pixel' (videotransformer.py:161-185) with injected uniforms.

    The reference draws with torch.multinomial (device-specific RNG stream); parity is therefore
    on the per-channel probabilities given the same previously drawn codes.
    uniforms: (b, nc).  Returns (codes (b, nc) int64, probs (b, nc, nv)).
    """
    pre = "ch_predictor."
    ti, hi, wi = pixel
    y = layer_norm(yl[:, :, ti, hi, wi], p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])
    b = y.size(0)
    nc = uniforms.size(1)
    onehot = torch.zeros(b, nc, nv)
    probs = []
    for k in range(nc):
        inp = y if k == 0 else torch.cat((y, onehot[:, :k].reshape(b, k * nv)), dim=1)
        u = F.linear(inp, p[pre + "U.%d.weight" % k], p[pre + "U.%d.bias" % k])
        o = F.linear(torch.relu(u), p[pre + "P.%d.weight" % k], p[pre + "P.%d.bias" % k])
        pr = torch.softmax(o / temp, 1)
        probs.append(pr)
        s = multinomial_from_uniform(pr, uniforms[:, k])
        onehot[torch.arange(b), k, s] = 1
    return onehot.argmax(dim=2), torch.stack(probs, 1)


def video_transformer_logits(p, context, sl, slice_idx, blocks_e, blocks_d, stride, nv=512,
                             pad_value=-1, return_hidden=False, class_idx=None):
    """VideoTransformer.forward mode='logits' (videotransformer.py:231-239)."""
    zl = vt_encoder(p, context, slice_idx, blocks_e, stride, nv, pad_value, class_idx)
    yl = vt_decoder(p, sl, zl, blocks_d)
    pred = channel_predictor_logits(p, sl, yl, nv)
    return (pred, zl, yl) if return_hidden else pred


# --------------------------------------------------------------------------------------------
# A17 / A18  transformer meta-architecture     (meta_arch/vt.py:230-314)
# --------------------------------------------------------------------------------------------
def vt_supervised_loss(p, context, sl, slice_idx, ignore_mask, blocks_e, blocks_d, stride, nv=512,
                       ignore_index=-100, class_idx=None):
    """compute_supervised_loss (vt.py:301-314): mean_k CE(pred_k, target_k, ignore_index)."""
    target = sl.masked_fill(ignore_mask, ignore_index)
    pred = video_transformer_logits(p, context, sl, slice_idx, blocks_e, blocks_d, stride, nv, class_idx=class_idx)
    loss = 0
    for k in range(len(pred)):
        loss = loss + F.cross_entropy(pred[k], target[:, k], ignore_index=ignore_index)
    return loss / len(pred), pred


def vt_logits_for_entire_video(p, video_btchw, blocks_e, blocks_d, stride, kernel, nv=512,
                               pad_value=-1, class_idx=None):
    """calculate_logits_for_entire_video (vt.py:230-282): (B,T,nc,H,W) codes -> (B,nc,nv,T,H,W)."""
    video = video_btchw.transpose(1, 2).contiguous()
    B, nc, T, H, W = video.shape
    st, sh, sw = stride
    idx2abc, _ = subscale_order(st, sh, sw)
    t, h, w = T // st, H // sh, W // sw
    logits = torch.zeros(B, nc, nv, T, H, W)
    for si, (a, b, c) in enumerate(idx2abc):
        smask = slice_mask(a, b, c, st, sh, sw, T, H, W)
        sl = video.masked_select(smask).clone().view(B, nc, t, h, w)
        vmask = visible_abc_mask(a, b, c, st, sh, sw, T, H, W)
        ctx = ss_shift(video.masked_fill(~vmask, pad_value), a, b, c, st, sh, sw, T, H, W, *kernel,
                       pad_value=pad_value)
        pred = video_transformer_logits(p, ctx, sl, torch.full((B,), si, dtype=torch.long),
                                        blocks_e, blocks_d, stride, nv, pad_value, class_idx=class_idx)
        for k in range(nc):
            logits[:, k] = logits[:, k].masked_scatter(smask, pred[k].reshape(-1))
    return logits
