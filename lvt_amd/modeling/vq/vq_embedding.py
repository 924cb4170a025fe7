"""Product vector quantiser with EMA codebooks (reference: vidgen/modeling/vq/vq_embedding.py:9-99,
vq_utils.py:5-65).

State layout is the reference's (`ve.i.embedding.weight`, `ve.i.running_size`, `ve.i.running_sum`)
but the tensors are views into three flat device buffers (num, K, D) so that all `num` codebooks
are quantised, gathered and EMA-updated by single kernel launches.

Deliberate, documented deviations:
  * `running_sum` owns its storage.  The reference registers it as `weight.detach()`, which on CPU
    aliases the codebook; on a GPU `.to(device)` de-aliases.  GPU semantics are kept.
  * multi-GPU EMA statistics are summed with ONE all-reduce of a packed (num, K, D+1) buffer instead of
    2*num all_gather+sum round trips (vq_embedding.py:46-47, 53-54); the sum is the same.
  * that all-reduce is OFF the critical path: the decoder only needs z_q_st, which comes from the pre-update codebook, so
    `straight_through_cl(z, defer=True)` starts the collective asynchronously right after the statistics kernel and returns;
    `finish_ema()` joins it, applies the EMA update and gathers z_q_bar for the commitment loss -- after the decoder forward
    has been enqueued (meta_arch/vqvae.py).  Same arithmetic, same order of updates as the reference (vq_embedding.py:46-59).
"""
import torch
from torch import nn

from ...hip import binding as L
from ...hip import ew, vq
from ...layers.all_reduce import all_reduce_sum_async_
from .. import convstack


class VQEmbedding(nn.Module):
    """One codebook: parameter / buffer container (vq_embedding.py:9-21)."""

    def __init__(self, K, D, ema):
        super().__init__()
        self.embedding = nn.Embedding(K, D)
        self.embedding.weight.data.uniform_(-1.0 / K, 1.0 / K)
        self.K, self.D, self.ema = K, D, ema
        if ema:
            self.eps, self.decay = 1e-5, 0.99
            self.register_buffer("running_size", torch.zeros(K))
            self.register_buffer("running_sum", self.embedding.weight.detach().clone())


class _StraightThroughFn(torch.autograd.Function):
    """vq_st as an autograd node (vq_utils.py:34-65): nearest code + gather from the PRE-update codebook; with EMA the
    assignment statistics are accumulated and their all-reduce is STARTED here (joined by DVQEmbedding.finish_ema)."""

    @staticmethod
    def forward(ctx, z2d, owner, P):
        w, rsize, rsum = owner._flat()
        idx = vq.nearest(z2d, w, P)
        z_q_st = vq.gather(idx, w)                       # from the PRE-update codebook
        stats = work = None
        if owner.ema:
            stats = vq.ema_accumulate(idx, z2d, owner.K)
            work = all_reduce_sum_async_(stats)
        owner._pending = (idx, stats, work)
        ctx.mark_non_differentiable(idx)
        return z_q_st, idx

    @staticmethod
    def backward(ctx, g_st, g_idx):
        return g_st, None, None                          # straight-through (vq_utils.py:52-54)


class _GatherWithGradFn(torch.autograd.Function):
    """rows of the codebooks selected by idx, differentiable w.r.t. the codebooks (CODEBOOK.EMA False): the backward pass is
    `grad_codebook.index_add_(0, indices, grad_rows)` (vq_utils.py:56-63) per codebook -- here the per-code row sums of the
    EMA-statistics kernel (fixed summation order, no atomics) applied to the gradient rows."""

    @staticmethod
    def forward(ctx, idx, owner, *weights):
        ctx.save_for_backward(idx)
        ctx.K, ctx.num, ctx.d = owner.K, owner.num, owner.D // owner.num
        return vq.gather(idx, owner._flat()[0])

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        stats = vq.ema_accumulate(idx, g.contiguous(), ctx.K)                 # (num, K, d + 1): sums | counts
        return (None, None) + tuple(stats[i, :, :ctx.d].contiguous() for i in range(ctx.num))


class _SingleStraightThroughFn(torch.autograd.Function):
    """vq_st of ONE wide codebook (vq_utils.py:34-65, vq_embedding.py:35-38)."""

    @staticmethod
    def forward(ctx, z2d, owner, P):
        w = owner.embedding.weight.detach()
        idx = vq.nearest_single(z2d, w)                          # (rows,)
        idx_g = owner._group_idx(idx, P)
        z_q_st = vq.gather(idx_g, owner._grouped(w))             # from the PRE-update codebook
        stats = work = None
        if owner.ema:
            stats = vq.ema_accumulate(idx_g, z2d, owner.K)       # (g, K, 65): the 64-d slices of the sums | the counts
            work = all_reduce_sum_async_(stats)
        owner._pending = (idx, idx_g, stats, work)
        ctx.mark_non_differentiable(idx)
        return z_q_st, idx

    @staticmethod
    def backward(ctx, g_st, g_idx):
        return g_st, None, None


class _SingleGatherWithGradFn(torch.autograd.Function):
    """rows of the codebook selected by idx, differentiable w.r.t. the codebook (EMA False: vq_embedding.py:61-64, vq_utils.py:56-63)."""

    @staticmethod
    def forward(ctx, idx_g, owner, weight):
        ctx.save_for_backward(idx_g)
        ctx.owner = owner
        return vq.gather(idx_g, owner._grouped(weight.detach()))

    @staticmethod
    def backward(ctx, g):
        (idx_g,) = ctx.saved_tensors
        o = ctx.owner
        stats = vq.ema_accumulate(idx_g, g.contiguous(), o.K)                      # (g, K, 65)
        return None, None, stats[:, :, :64].permute(1, 0, 2).reshape(o.K, o.D).contiguous()


class SingleVQEmbedding(VQEmbedding):
    """`VQEmbedding` used directly as the quantiser: CODEBOOK.NUM == 1, the default of the config tree (config/defaults.py:79,
    meta_arch/vqvae.py:25-27).  Same parameters, buffers and state_dict keys as the reference's class (`embedding.weight`,
    `running_size`, `running_sum`); latents carry no codebook axis: (N, H, W).

    The search runs as engine GEMM + lvt_vq_argmax_scores (hip/vq.py nearest_single).  Gather and the EMA statistics reuse the
    product quantiser's kernels, which work on 64-d groups: the (K, D) codebook is viewed as D / 64 groups that all select with
    the SAME index -- group g of code k is e_k[64 g : 64 g + 64] -- so the gathered rows, the per-code sums and the counts (equal
    in every group) are exactly those of the wide codebook, and lvt_vq_ema_finalize applies the reference's update
    (vq_embedding.py:48-59) to every group with the same counts."""

    def __init__(self, K, D, ema):
        super().__init__(K, D, ema)
        if D % 64:
            raise NotImplementedError("the HIP quantiser works on 64-d groups: CODEBOOK.DIM must be a multiple of 64 (got %d)" % D)
        self.num, self.groups = 1, D // 64
        self._pending = None

    # ---- 64-d group views ------------------------------------------------------------------------------------------------
    def _grouped(self, t):
        """(K, D) -> (g, K, 64) contiguous; (K,) -> (g, K)."""
        if t.dim() == 1:
            return t.unsqueeze(0).expand(self.groups, self.K).contiguous()
        out = t.view(self.K, self.groups, 64).permute(1, 0, 2).contiguous()
        if L.f16x2() and L._valid_amax(t) is not None:
            L.set_amax(out, L._valid_amax(t))
        return out

    def _ungrouped(self, t):
        return t.permute(1, 0, 2).reshape(self.K, self.D)

    def _group_idx(self, idx, P):
        """(rows,) -> (rows / P, g, P): every group selects with the same index."""
        return idx.view(-1, 1, P).expand(-1, self.groups, P).contiguous()

    def _flat(self):
        """The tensors a data-parallel run broadcasts from rank 0 (meta_arch/vqvae.py wrap_parallel)."""
        return (self.embedding.weight.data, getattr(self, "running_size", None), getattr(self, "running_sum", None))

    # ---- channels-last fast paths (the interface VQVAEModel uses: see DVQEmbedding) ---------------------------------------
    def indices_cl(self, z_cl):
        n, _, h, w, d = z_cl.shape
        L.require(z_cl)
        return vq.nearest_single(z_cl.view(n * h * w, d), self.embedding.weight).view(n, h, w)

    def straight_through_cl(self, z_cl, defer=False):
        n, _, h, w, d = z_cl.shape
        L.require(z_cl)
        if self._pending is not None:
            raise L.LvtError("SingleVQEmbedding: the EMA update of the previous pass was never finished (finish_ema)")
        z_q_st, idx = _SingleStraightThroughFn.apply(z_cl.view(n * h * w, d), self, h * w)
        self.last_indices = idx.view(n, h, w)
        self._pending_shape = (n, 1, h, w, d)
        z_q_st = z_q_st.view(n, 1, h, w, d)
        return z_q_st if defer else (z_q_st, self.finish_ema())

    def finish_ema(self):
        if self._pending is None:
            raise L.LvtError("SingleVQEmbedding.finish_ema without a pending straight_through_cl")
        idx, idx_g, stats, work = self._pending
        self._pending = None
        w = self.embedding.weight
        if stats is not None:
            if work is not None:
                work.wait()
            wg, rs, rsum = self._grouped(w.data), self._grouped(self.running_size), self._grouped(self.running_sum)
            vq.ema_finalize(stats, rs, rsum, wg, self.decay, self.eps)
            with torch.no_grad():
                w.data.copy_(self._ungrouped(wg))
                self.running_size.copy_(rs[0])
                self.running_sum.copy_(self._ungrouped(rsum))
            L.drop_amax(w)
            return vq.gather(idx_g, wg).view(*self._pending_shape)               # from the POST-update codebook
        return _SingleGatherWithGradFn.apply(idx_g, self, w).view(*self._pending_shape)

    def abandon_ema(self):
        pend, self._pending = self._pending, None
        if pend is not None and pend[3] is not None:
            pend[3].wait()

    def embed_cl(self, latents):
        """(N, H, W) int64 -> (N, 1, H, W, D) channels-last."""
        n, h, w = latents.shape
        L.require(latents)
        out = vq.gather(self._group_idx(latents.contiguous().view(-1), h * w), self._grouped(self.embedding.weight.detach()))
        return out.view(n, 1, h, w, self.D)

    # ---- the reference's call contract (vq_embedding.py:23-33) -------------------------------------------------------------
    def forward(self, z_e_x, mode=""):
        if mode == "":
            return self.indices_cl(convstack._LayoutIn.apply(z_e_x))
        if mode == "st":
            st, bar = self.straight_through_cl(convstack._LayoutIn.apply(z_e_x))
            return convstack._LayoutOut.apply(st, self.D), convstack._LayoutOut.apply(bar, self.D)
        if mode == "emb":
            return self.embed_cl(z_e_x).squeeze(1)           # (N, H, W, D), as nn.Embedding yields
        raise ValueError


class DVQEmbedding(nn.Module):
    def __init__(self, num, K, D, ema):
        super().__init__()
        assert D % num == 0
        if D // num != 64:
            raise NotImplementedError("the HIP quantiser is instantiated for 64-d sub-vectors (got %d)" % (D // num))
        self.num, self.D, self.K, self.ema = num, D, K, ema
        self.decay, self.eps = 0.99, 1e-5
        self.ve = nn.ModuleList([VQEmbedding(K, D // num, ema) for _ in range(num)])
        self._flat_w = self._flat_rs = self._flat_rsum = None

    # ---- flat storage ----------------------------------------------------------------------------
    def _flat(self):
        """(weights (num,K,d), running_size (num,K), running_sum (num,K,d)) whose slices ARE the
        per-codebook parameters / buffers.  Rebuilt whenever a `.to()` / load replaced the storage."""
        w0 = self.ve[0].embedding.weight
        fw = self._flat_w
        ok = fw is not None and fw.device == w0.device and all(
            self.ve[i].embedding.weight.data_ptr() == fw[i].data_ptr()
            and (not self.ema or (self.ve[i].running_size.data_ptr() == self._flat_rs[i].data_ptr()
                                  and self.ve[i].running_sum.data_ptr() == self._flat_rsum[i].data_ptr()))
            for i in range(self.num))
        if not ok:
            with torch.no_grad():
                fw = torch.stack([v.embedding.weight.data for v in self.ve]).contiguous()
                frs = frsum = None
                if self.ema:                 # (a trained codebook, CODEBOOK.EMA False, has no running buffers: vq_embedding.py:17-21)
                    frs = torch.stack([v.running_size for v in self.ve]).contiguous()
                    frsum = torch.stack([v.running_sum for v in self.ve]).contiguous()
                for i, v in enumerate(self.ve):
                    v.embedding.weight.data = fw[i]
                    if self.ema:
                        v.running_size = frs[i]
                        v.running_sum = frsum[i]
            self._flat_w, self._flat_rs, self._flat_rsum = fw, frs, frsum
        return self._flat_w, self._flat_rs, self._flat_rsum

    # ---- channels-last fast paths (used by VQVAEModel) -----------------------------------------------
    def indices_cl(self, z_cl):
        """(N,1,H,W,D) -> (N,num,H,W) int64."""
        n, _, h, w, d = z_cl.shape
        L.require(z_cl)
        idx = vq.nearest(z_cl.view(n * h * w, d), self._flat()[0], h * w)
        return idx.view(n, self.num, h, w)

    def straight_through_cl(self, z_cl, defer=False):
        """(N,1,H,W,D) -> (z_q_st, z_q_bar) channels-last; updates the EMA state (vq_embedding.py:35-66).
        defer=True returns z_q_st only and leaves the EMA update + z_q_bar to `finish_ema()`: the statistics all-reduce of a
        data-parallel run then overlaps whatever the caller enqueues in between (the decoder forward)."""
        n, _, h, w, d = z_cl.shape
        L.require(z_cl)
        if getattr(self, "_pending", None) is not None:
            raise L.LvtError("DVQEmbedding: the EMA update of the previous pass was never finished (finish_ema)")
        z_q_st, idx = _StraightThroughFn.apply(z_cl.view(n * h * w, d), self, h * w)
        self.last_indices = idx.view(n, self.num, h, w)      # kept for evaluators / tests
        self._pending_shape = (n, 1, h, w, d)
        z_q_st = z_q_st.view(n, 1, h, w, d)
        return z_q_st if defer else (z_q_st, self.finish_ema())

    def finish_ema(self):
        """Join the statistics all-reduce, apply the EMA update (vq_embedding.py:48-59) and return z_q_bar, the codes of this
        pass gathered from the UPDATED codebook (channels-last)."""
        pend = getattr(self, "_pending", None)
        if pend is None:
            raise L.LvtError("DVQEmbedding.finish_ema without a pending straight_through_cl")
        idx, stats, work = pend
        self._pending = None
        w, rsize, rsum = self._flat()
        if stats is not None:
            if work is not None:
                work.wait()
            vq.ema_finalize(stats, rsize, rsum, w, self.decay, self.eps)
        if not self.ema:
            # a trained codebook: z_q_bar carries the graph to the `num` weight parameters (vq_embedding.py:61-64); their
            # gradient is the reference's index_add_ of the incoming rows (vq_utils.py:56-63)
            return _GatherWithGradFn.apply(idx, self, *[v.embedding.weight for v in self.ve]).view(*self._pending_shape)
        return vq.gather(idx, w).view(*self._pending_shape)      # from the POST-update codebook

    def abandon_ema(self):
        """Error path of a deferred pass: wait for the statistics all-reduce that was started and forget the pending update
        (the codebook stays as it was before the pass)."""
        pend = getattr(self, "_pending", None)
        if pend is not None:
            self._pending = None
            if pend[2] is not None:
                pend[2].wait()

    def embed_cl(self, latents):
        """(N,num,H,W) int64 -> (N,1,H,W,D) channels-last."""
        n, num, h, w = latents.shape
        L.require(latents)
        out = vq.gather(latents.contiguous().view(n, num, h * w), self._flat()[0])
        return out.view(n, 1, h, w, self.D)

    # ---- the reference's call contract (vq_embedding.py:77-99) ------------------------------------
    def forward(self, z_e_x, mode=""):
        if mode == "":
            assert z_e_x.dim() == 4
            return self.indices_cl(convstack._LayoutIn.apply(z_e_x))
        if mode == "st":
            assert z_e_x.dim() == 4
            st, bar = self.straight_through_cl(convstack._LayoutIn.apply(z_e_x))
            return convstack._LayoutOut.apply(st, self.D), convstack._LayoutOut.apply(bar, self.D)
        if mode == "emb":
            return self.embed_cl(z_e_x).squeeze(1)       # (N,H,W,D), as torch.cat(..., dim=-1) yields
        raise ValueError
