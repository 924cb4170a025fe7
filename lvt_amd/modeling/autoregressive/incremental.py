"""Incremental (KV-cache) decoding of one subscale slice.

The reference samples a slice by re-running the whole 8-layer causal decoder over all S tokens for every
generated pixel (meta_arch/vt.py:121-131): S full passes per slice.  Because the decoder front end is
strictly causal (MaskedConv3d) and every attention layer is causally masked, the hidden state of token i
depends only on tokens < i, so token i can be computed alone against cached keys / values of the earlier
tokens: S single-token steps per slice, the same arithmetic per row (the masked columns of the full pass
carry exactly zero probability).  `tests/test_gpu_sampling.py` checks step(i) against row i of the full pass.

The position being decoded lives in DEVICE memory (`IncrementalDecoder.pos`, one int32): every kernel of a step
that addresses by position (neighbour gather, the row of the K/V caches written by the q/k/v product, the residual
row of the front end, the number of keys of the decode attention, the write-back of the drawn codes) reads it
there, and the last launch of a step increments it.  The launch arguments of a step are therefore the same for
every position, and ONE captured hipGraph per group of videos (two when a slice mixes primed and generated
positions) is replayed for all S positions of all slices -- not one graph per position.  Everything a captured
launch touches is owned by the decoder object of its group and allocated before the capture; nothing in a graph
points into `binding.workspace()` (which refuses to serve a capturing stream).
"""
import math
import os

import torch

from ...hip import binding as L
from ...hip import ew, tx
from ...hip import gemm as G


class IncrementalDecoder:
    def __init__(self, decoder, zl_tok, b, thw):
        t, h, w = thw
        self.dec, self.b, self.thw, self.S = decoder, b, thw, t * h * w
        dev = zl_tok.device
        rows = b * self.S
        d = decoder.linear_projector.weight.shape[0]
        self.d = d
        self.base = torch.empty(rows, d, dtype=torch.float32, device=dev)
        cw = decoder.conv.conv.weight
        self.de = cw.shape[1]
        self.nc = len(decoder.ch_embedder)
        # codes of the slice being decoded, with one extra always-padded position that out-of-range causal
        # neighbours point at; `self.sl` is the (b, nc, t, h, w) view callers write drawn codes into
        self.sl_ext = torch.full((b, self.nc, self.S + 1), -1, dtype=torch.int64, device=dev)
        self.sl = self.sl_ext[:, :, :self.S].view(b, self.nc, t, h, w)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)       # the device-side cursor (module docstring)
        # single-position causal conv: x_i = sum over the taps that can see data of W_tap . emb[neighbour_tap(i)]
        kt, kh, kw = cw.shape[2:]
        self.taps = [(jt, jh, jw) for jt in range(kt) for jh in range(kh) for jw in range(kw)
                     if not (jt == kt - 1 and jh == kh - 1 and jw >= kw // 2 and kw // 2 > 0)      # zeroed (non-causal) taps
                     and kt - 1 - jt < t and kh - 1 - jh < h and abs(jw - kw // 2) < w]            # never inside the volume
        nb = torch.full((self.S, len(self.taps)), self.S, dtype=torch.int64)
        for ti in range(t):
            for hi in range(h):
                for wi in range(w):
                    for j, (jt, jh, jw) in enumerate(self.taps):
                        tt, hh, ww = ti + jt - (kt - 1), hi + jh - (kh - 1), wi + jw - kw // 2
                        if tt >= 0 and hh >= 0 and 0 <= ww < w:
                            nb[(ti * h + hi) * w + wi, j] = (tt * h + hh) * w + ww
        self.nb = nb.to(dev)
        self.wfront = torch.empty(d, len(self.taps) * self.de, dtype=torch.float32, device=dev)
        self.tables = torch.empty(self.nc * decoder.ch_embedder[0].weight.shape[0], self.de, dtype=torch.float32, device=dev)
        self.layers = list(decoder.block_local_attention)
        m0 = self.layers[0].mha
        self.na, self.da = m0.na, m0.da
        hd = self.na * self.da
        # per layer one (3, b, S, hd) tensor: slot 0 holds the queries, 1 the key cache, 2 the value cache, so that one
        # batched small-M GEMM (batch 3, uniform weight / output strides) writes q_i, k_i and v_i of every sample
        self.qkv = [torch.zeros(3, b, self.S, hd, dtype=torch.float32, device=dev) for _ in self.layers]
        self.kc = [t[1] for t in self.qkv]
        self.vc = [t[2] for t in self.qkv]
        # per-head projection weights (na, d, da) re-laid out as one k-contiguous (na*da, d) matrix per projection,
        # the layout the matrix-core small-M GEMM streams; refreshed at every begin_slice (weights may have been
        # trained in between), in place so that captured graphs keep reading the same buffers
        self.wqkv = [torch.empty(3, hd, d, dtype=torch.float32, device=dev) for _ in self.layers]
        # split-K partial tiles of the two residual products of a layer (see step): sized here, outside any capture
        self._pbuf = {}
        self._partials("proj", max(2, hd // 128))
        dff = self.layers[0].ffn[1].weight.shape[0]
        self._partials("ffn", max(2, dff // 128))
        # scratch of the generic products that split k themselves (reductions of 1024 and more, hip/gemm.py)
        kmax = max(hd, dff, d, len(self.taps) * self.de)
        self._split_ws(4 * b * max(d, dff, hd) * max(1, kmax // 256))
        self.begin_slice(zl_tok)

    def _refresh_weights(self):
        self.dec.conv.rezero_()
        cw = self.dec.conv.conv.weight.detach()
        for j, (jt, jh, jw) in enumerate(self.taps):
            self.wfront[:, j * self.de:(j + 1) * self.de].copy_(cw[:, :, jt, jh, jw])
        torch.cat([e.weight.detach() for e in self.dec.ch_embedder], dim=0, out=self.tables)
        for layer, bufs in zip(self.layers, self.wqkv):
            m = layer.mha
            for w, buf in zip((m.w_q, m.w_k, m.w_v), bufs):
                buf.view(self.na, self.da, self.d).copy_(w.detach().transpose(1, 2))

    def begin_slice(self, zl_tok):
        """position signal + projection of the encoder output: fixed for the whole slice (written in place so
        that captured graphs keep seeing the same buffer).  Stale cache rows need no reset: step(i) only reads
        keys 0..i, all of which are rewritten while the new slice is decoded."""
        t, h, w = self.thw
        self._refresh_weights()
        G.gemm(zl_tok, self.dec.linear_projector.weight, self.base, self.b * self.S, self.d, self.d)
        self.dec.positional_encoder.add_tokens_(self.base, t, h, w)

    def _front_row(self):
        """x_i = causal_conv(sum_k Emb_k(slice))[i] + pos[i] + proj(zl)[i]  -> (b, d) for i = the device cursor: the codes
        of the causal neighbours of position i are gathered (integer plumbing), embedded and contracted with the packed taps."""
        b, nc, nt = self.b, self.nc, len(self.taps)
        nv = self.tables.shape[0] // nc
        codes = tx.decode_gather_codes(self.sl_ext.view(b * nc, self.S + 1), self.nb, self.pos)   # (b*nc, taps), -1 = outside
        a = tx.embbag_fwd(codes, nc * nt, nt, b * nt, [k * nt for k in range(nc)], [k * nv for k in range(nc)],
                          self.tables, self.de)                                      # (b*taps, de) == (b, taps*de)
        x = torch.empty(b, self.d, dtype=torch.float32, device=a.device)
        G.gemm_small(a, self.wfront, x, b, self.d, nt * self.de, flags=L.EPI_BIAS | L.EPI_RESIDUAL,
                     bias=self.dec.conv.conv.bias, res=self.base, ldr=self.S * self.d, split_ws=self._split_ws,
                     pos=self.pos, r_pos=self.d)
        return x

    def _partials(self, which, splits):
        """Split-K partial buffers of the decode step: ordinary tensors owned by this object (sized in __init__), so
        captured launches keep pointing at live memory."""
        buf = self._pbuf.get(which)
        if buf is None or buf.numel() < splits * self.b * self.d:
            buf = self._pbuf[which] = torch.empty(splits * self.b * self.d, dtype=torch.float32, device=self.base.device)
        return buf

    def _split_ws(self, nbytes):
        """Split-K scratch of the generic small-M products of a step, owned by this decoder (one per hipGraph group)."""
        buf = self._pbuf.get("generic")
        if buf is None or buf.numel() * 4 < nbytes:
            buf = self._pbuf["generic"] = torch.empty((int(nbytes) + 3) // 4, dtype=torch.float32, device=self.base.device)
        return buf

    def step(self, sl, i):
        """Hidden state y_i (b, d) of token i given the codes of tokens < i in `sl`; fills the caches at i.
        (Eager entry: points the device cursor at i; the cursor is not advanced.)"""
        if not 0 <= i < self.S:
            raise L.LvtError("IncrementalDecoder.step: position %d outside the slice of %d tokens" % (i, self.S))
        if sl.data_ptr() != self.sl.data_ptr():
            self.sl.copy_(sl)
        self.pos.fill_(i)
        return self.step_at_cursor()

    def step_at_cursor(self):
        """y (b, d) of the token the device cursor points at; every position-dependent address is formed on the device."""
        b, d, S = self.b, self.d, self.S
        na, da = self.na, self.da
        hd = na * da
        x = self._front_row()
        dev = x.device
        # Both products of a layer that end in a residual (output projection, FFN down-projection) are followed by a
        # LayerNorm: they run split-K (one 128-deep chunk per workgroup, 4x the workgroups) and leave raw partial
        # tiles; the LayerNorm launch sums them, adds bias / residual and normalises (`pend`: partials not reduced yet).
        KS = 128
        nlayers = len(self.layers)
        pend = None
        for li, layer in enumerate(self.layers):
            m, f = layer.mha, layer.ffn
            if pend is None:
                xn, _, _ = ew.layernorm_fwd(x, m.layer_norm.weight, m.layer_norm.bias, save_stats=False)
            else:
                x, xn = G.splitsum_layernorm(pend[0], pend[1], b, d, m.layer_norm.weight, m.layer_norm.bias,
                                             bias=pend[2], res=pend[3])
            # q_i / k_i / v_i of every sample in one launch: output row i of slot z, row stride S*hd, slot stride b*S*hd
            qkv = self.qkv[li]
            G.gemm_small(xn, self.wqkv[li], qkv, b, hd, d, ldc=S * hd, batch=3, sB=hd * d, sC=b * S * hd,
                         split_ws=self._split_ws, pos=self.pos, c_pos=hd)
            o = tx.attn_decode(qkv, self.kc[li], self.vc[li], na, 0, math.sqrt(da), layer.dt_bank,
                               layer.dh_bank, layer.dw_bank, layer.block_size, ldq=S * hd, pos=self.pos, q_pos=hd)
            if hd % KS == 0 and d % 4 == 0:
                ws = G.gemm_small_partial(o, m.proj.weight, b, d, hd, hd // KS, self._partials("proj", hd // KS))
                y1, fn = G.splitsum_layernorm(ws, hd // KS, b, d, f[0].weight, f[0].bias, res=x)
            else:
                y1 = torch.empty(b, d, dtype=torch.float32, device=dev)
                G.gemm_small(o, m.proj.weight, y1, b, d, hd, flags=L.EPI_RESIDUAL, res=x, split_ws=self._split_ws)
                fn, _, _ = ew.layernorm_fwd(y1, f[0].weight, f[0].bias, save_stats=False)
            dff = f[1].weight.shape[0]
            h1 = torch.empty(b, dff, dtype=torch.float32, device=dev)
            G.gemm_small(fn, f[1].weight, h1, b, dff, d, flags=L.EPI_BIAS | L.EPI_RELU, bias=f[1].bias, split_ws=self._split_ws)
            if li + 1 < nlayers and dff % KS == 0 and dff // KS >= 2 and d % 4 == 0:
                ws = G.gemm_small_partial(h1, f[3].weight, b, d, dff, dff // KS, self._partials("ffn", dff // KS))
                pend = (ws, dff // KS, f[3].bias, y1)
            else:
                pend = None
                x = torch.empty(b, d, dtype=torch.float32, device=dev)
                G.gemm_small(h1, f[3].weight, x, b, d, dff, flags=L.EPI_BIAS | L.EPI_RESIDUAL, bias=f[3].bias, res=y1, split_ws=self._split_ws)
        return x


class GraphedSliceSampler:
    """One decoding step (decoder step + channel-predictor draw + write-back of the drawn codes + cursor increment) as a
    replayed hipGraph.

    A step is ~110 tiny launches (M = batch rows); eagerly it is bound by host launch overhead (~3.7 ms/step measured at
    16 videos).  Because the position is a device-side cursor the launch sequence AND its arguments are identical for
    every position: the first step of each flavour (drawing / cache-filling only) runs eagerly and is then captured once;
    every later position of every slice replays that graph.  LVT_DECODE_GRAPHS=0 keeps every step eager."""

    def __init__(self, vt_module, b, thw, temp=1.0):
        self.vt, self.b, self.thw, self.temp = vt_module, b, thw, temp
        self.dec = None
        self.sl = None
        self.graphs = {}
        self.uniforms = None
        self.enabled = os.environ.get("LVT_DECODE_GRAPHS", "1") != "0"
        self._next = 0                                      # host mirror of the device cursor (argument checking only)

    def begin_slice(self, zl_tok, sl):
        if self.dec is None:
            self.dec = IncrementalDecoder(self.vt.decoder, zl_tok, self.b, self.thw)
            self.sl = self.dec.sl                          # drawn codes go straight into the decoder's buffer
        else:
            self.dec.begin_slice(zl_tok)
        self.vt.ch_predictor.prepare_decode()
        self.sl.copy_(sl)
        self.dec.pos.zero_()
        self._next = 0
        # every uniform of the slice in one draw (S, nc, b); the steps index it with the device cursor
        if self.uniforms is None:
            self.uniforms = torch.empty(self.dec.S, self.dec.nc, self.b, dtype=torch.float32, device=self.sl.device)
        self.uniforms.uniform_()

    def _body(self, sample):
        dec = self.dec
        y = dec.step_at_cursor()
        drawn = self.vt.ch_predictor.sample_from_rows(y, self.temp, uniforms=self.uniforms, pos=dec.pos,
                                                      split_ws=dec._split_ws).reshape(-1) if sample else None     # (b*nc,)
        tx.decode_commit(dec.sl_ext.view(dec.b * dec.nc, dec.S + 1), dec.pos, drawn)

    def step(self, pos, sample):
        """Decode position `pos` (must be the next one: positions of a slice are visited in order)."""
        if pos != self._next:
            raise L.LvtError("GraphedSliceSampler.step(%d): the device cursor is at %d" % (pos, self._next))
        if pos >= self.dec.S:
            raise L.LvtError("GraphedSliceSampler.step(%d): the slice has %d positions (a replay past the end would address "
                             "beyond the caches)" % (pos, self.dec.S))
        self._next += 1
        g = self.graphs.get(bool(sample))
        if g is not None:
            g.replay()
            return
        self._body(sample)                                # eager (also the warm-up the capture needs)
        if not self.enabled:
            return
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):                         # recorded, not executed: the cursor keeps its value
            self._body(sample)
        self.graphs[bool(sample)] = g
