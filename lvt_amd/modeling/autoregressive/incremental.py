"""Incremental (KV-cache) decoding of one subscale slice.

The reference samples a slice by re-running the whole 8-layer causal decoder over all S tokens for every
generated pixel (meta_arch/vt.py:121-131): S full passes per slice.  Because the decoder front end is
strictly causal (MaskedConv3d) and every attention layer is causally masked, the hidden state of token i
depends only on tokens < i, so token i can be computed alone against cached keys / values of the earlier
tokens: S single-token steps per slice, the same arithmetic per row (the masked columns of the full pass
carry exactly zero probability).  `tests/test_gpu_sampling.py` checks step(i) against row i of the full pass.
"""
import math

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
        self.begin_slice(zl_tok)
        decoder.conv.rezero_()
        cw = decoder.conv.conv.weight
        self.de = cw.shape[1]
        kt, kh, kw = cw.shape[2:]
        self.geom = G.conv_geom(b, t, h, w, self.de, d, (kt, kh, kw), (1, 1, 1), (kt - 1, kh - 1, kw // 2), out=(t, h, w))
        self.wp = G.pack_weight(self.geom, cw, self.de, d)
        self.tables = torch.cat([e.weight for e in decoder.ch_embedder], dim=0).contiguous()
        self.layers = list(decoder.block_local_attention)
        m0 = self.layers[0].mha
        self.na, self.da = m0.na, m0.da
        hd = self.na * self.da
        self.kc = [torch.zeros(b, self.S, hd, dtype=torch.float32, device=dev) for _ in self.layers]
        self.vc = [torch.zeros(b, self.S, hd, dtype=torch.float32, device=dev) for _ in self.layers]

    def begin_slice(self, zl_tok):
        """position signal + projection of the encoder output: fixed for the whole slice (written in place so
        that captured graphs keep seeing the same buffer).  Stale cache rows need no reset: step(i) only reads
        keys 0..i, all of which are rewritten while the new slice is decoded."""
        t, h, w = self.thw
        G.gemm(zl_tok, self.dec.linear_projector.weight, self.base, self.b * self.S, self.d, self.d)
        self.dec.positional_encoder.add_tokens_(self.base, t, h, w)

    def _front_row(self, sl, i):
        """x_i = causal_conv(sum_k Emb_k(slice))[i] + pos[i] + proj(zl)[i]  -> (b, d)."""
        b, nc = sl.shape[:2]
        S, nv = self.S, self.tables.shape[0] // nc
        emb = tx.embbag_fwd(sl, nc * S, S, b * S, [k * S for k in range(nc)], [k * nv for k in range(nc)],
                            self.tables, self.de)
        t, h, w = self.thw
        x = G.conv_fwd(self.geom, emb.view(b, t, h, w, self.de), self.wp, bias=self.dec.conv.conv.bias,
                       res=self.base.view(b, t, h, w, self.d))
        return x.view(b, S, self.d)[:, i].contiguous()

    def step(self, sl, i):
        """Hidden state y_i (b, d) of token i given the codes of tokens < i in `sl`; fills the caches at i."""
        b, d, S = self.b, self.d, self.S
        na, da = self.na, self.da
        hd = na * da
        x = self._front_row(sl.contiguous(), i)
        dev = x.device
        for li, layer in enumerate(self.layers):
            m, f = layer.mha, layer.ffn
            xn, _, _ = ew.layernorm_fwd(x, m.layer_norm.weight, m.layer_norm.bias, save_stats=False)
            q = torch.empty(b, hd, dtype=torch.float32, device=dev)
            G.gemm_small(xn, m.w_q, q, b, da, d, tb=1, lda=d, ldb=da, ldc=hd, batch=na, sB=d * da, sC=da)
            for w_, cache in ((m.w_k, self.kc[li]), (m.w_v, self.vc[li])):
                # write row i of every sample straight into the cache: C = cache[0, i], row stride S*hd
                G.gemm_small(xn, w_, cache.view(-1)[i * hd:], b, da, d, tb=1, lda=d, ldb=da, ldc=S * hd, batch=na,
                             sB=d * da, sC=da)
            o = tx.attn_decode(q, self.kc[li], self.vc[li], na, i, math.sqrt(da), layer.dt_bank, layer.dh_bank,
                               layer.dw_bank, layer.block_size)
            y1 = torch.empty(b, d, dtype=torch.float32, device=dev)
            G.gemm_small(o, m.proj.weight, y1, b, d, hd, flags=L.EPI_RESIDUAL, res=x)
            fn, _, _ = ew.layernorm_fwd(y1, f[0].weight, f[0].bias, save_stats=False)
            h1 = torch.empty(b, f[1].weight.shape[0], dtype=torch.float32, device=dev)
            G.gemm_small(fn, f[1].weight, h1, b, f[1].weight.shape[0], d, flags=L.EPI_BIAS | L.EPI_RELU, bias=f[1].bias)
            x = torch.empty(b, d, dtype=torch.float32, device=dev)
            G.gemm_small(h1, f[3].weight, x, b, d, f[3].weight.shape[1], flags=L.EPI_BIAS | L.EPI_RESIDUAL,
                         bias=f[3].bias, res=y1)
        return x


class GraphedSliceSampler:
    """Per-position hipGraphs of (decoder step + channel-predictor draw + write-back of the drawn codes).

    One decoding step is ~110 tiny launches (M = batch rows); eagerly it is bound by host launch overhead
    (~3.7 ms/step measured at 16 videos).  Every position of a slice runs the same launch sequence on the same
    buffers, so the sequence is captured once per position (first slice: eager + capture) and replayed for
    every later slice / video batch."""

    def __init__(self, vt_module, b, thw, temp=1.0):
        self.vt, self.b, self.thw, self.temp = vt_module, b, thw, temp
        self.dec = None
        self.sl = None
        self.graphs = {}
        self.enabled = True

    def begin_slice(self, zl_tok, sl):
        if self.dec is None:
            self.dec = IncrementalDecoder(self.vt.decoder, zl_tok, self.b, self.thw)
            self.sl = sl.clone()
        else:
            self.dec.begin_slice(zl_tok)
            self.sl.copy_(sl)

    def _body(self, pos, sample):
        y = self.dec.step(self.sl, pos)
        if sample:
            t, h, w = self.thw
            ti, rem = divmod(pos, h * w)
            hi, wi = divmod(rem, w)
            self.sl[:, :, ti, hi, wi] = self.vt.ch_predictor.sample_from_rows(y, self.temp)

    def step(self, pos, sample):
        key = (pos, bool(sample))
        g = self.graphs.get(key)
        if g is not None:
            g.replay()
            return
        self._body(pos, sample)                           # eager (also the warm-up the capture needs)
        if not self.enabled:
            return
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._body(pos, sample)
            self.graphs[key] = g
        except Exception:                                 # capture unsupported in this environment: stay eager
            self.enabled = False
            self.graphs.clear()
            torch.cuda.synchronize()
