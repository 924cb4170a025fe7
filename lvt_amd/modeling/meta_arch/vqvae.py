"""VQ-VAE meta-architecture (reference: vidgen/modeling/meta_arch/vqvae.py:17-124).

supervised step:  x -> ResEncoder -> z_e -> DVQ "st" (indices, z_q_st from the pre-update codebook,
EMA update, z_q_bar from the post-update codebook) -> ResDecoder -> x_tilde;
losses: lambda * mse(x_tilde, x)  and  beta * mse(z_e, sg(z_q_bar)).
"""
import os

from ...solver import build_lr_scheduler, build_optimizer
from ...utils.checkpoint import Checkpointer
from ..loss import PixelLoss
from ..loss.loss import mse
from ..vq import DVQEmbedding, SingleVQEmbedding
from .ae import AutoEncoderModel
from .build import META_ARCH_REGISTRY


@META_ARCH_REGISTRY.register()
class VQVAEModel(AutoEncoderModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        cb = cfg.MODEL.CODEBOOK
        self.use_codebook_ema = cb.EMA
        if cb.NUM == 1:          # the default of the config tree: `VQEmbedding` used directly (vqvae.py:25-27)
            self.codebook = SingleVQEmbedding(cb.SIZE, cb.DIM, self.use_codebook_ema)
        else:
            self.codebook = DVQEmbedding(cb.NUM, cb.SIZE, cb.DIM, self.use_codebook_ema)
        if self.use_codebook_ema:
            self._set_requires_grad(self.codebook.parameters(), False)
        self.pixel_loss = PixelLoss(cfg)
        self.beta = cb.BETA
        self.to(self.device)

    def train(self, mode=True):
        super().train(mode)
        self.codebook.train(mode)
        return self

    def wrap_parallel(self, device_ids, broadcast_buffers):
        """Encoder / generator gradients are averaged by the bucketed reducer (see AutoEncoderModel).  The
        EMA codebook takes no gradient; its statistics are summed across ranks inside the quantiser.  The
        reference neither wraps nor synchronises the codebook, so with per-rank seeds (SEED + rank) the
        replicas start from different codebooks and only converge geometrically; here rank 0's codebook
        and EMA buffers are broadcast once so that all replicas stay bit-identical (documented deviation,
        single-GPU behaviour unaffected)."""
        import torch
        import torch.distributed as dist
        from ...utils import comm
        super().wrap_parallel(device_ids, broadcast_buffers)
        if not self.use_codebook_ema:
            # a trained codebook is a DDP module of its own in the reference (vqvae.py:46-49): its gradients are averaged too
            from ...engine.grad_reducer import BucketedGradReducer
            self._reducers.append(BucketedGradReducer(self.codebook.parameters()))
        if comm.get_world_size() > 1:
            with torch.no_grad():
                for t in self.codebook._flat():
                    if t is not None:
                        dist.broadcast(t, 0)

    def _generator_parameters(self):
        params = super()._generator_parameters()
        if not self.use_codebook_ema:
            params += list(self.codebook.parameters())
        return params

    def forward(self, data, mode="inference"):
        return super().forward(data, mode)

    # ---- channels-last internals --------------------------------------------------------------------
    def _latent_cl(self, x):
        return self.codebook.indices_cl(self.encoder.forward_cl(x))          # (N,num,h,w) int64

    def _latent_public(self, latent):
        return latent

    def _decode_cl(self, latent):
        return self.generator.forward_cl(self.codebook.embed_cl(latent))

    def _supervised_loss_cl(self, x, return_x=False):
        z_e = self.encoder.forward_cl(x)
        # the decoder needs z_q_st only (pre-update codebook): the all-reduce of the EMA statistics that the quantiser
        # started runs beside the decoder forward and is joined afterwards (reference order of updates: vq_embedding.py:46-59)
        z_q_st = self.codebook.straight_through_cl(z_e, defer=True)
        try:
            x_tilde = self.generator.forward_cl(z_q_st)
        except BaseException:
            self.codebook.abandon_ema()     # (a failed decoder pass must not leave the quantiser unusable: join + drop the update)
            raise
        z_q_bar = self.codebook.finish_ema()
        c = len(self.cfg.MODEL.PIXEL_MEAN)
        loss = {
            "loss_reconstruction": self.pixel_loss(x_tilde, x, denom=x.numel() // x.shape[-1] * c),
            "loss_commitment": mse(z_e, z_q_bar.detach(), scale=self.beta),
        }
        if not self.use_codebook_ema:
            # vector-quantisation objective of a TRAINED codebook, under the reference's key (vqvae.py:83-84)
            loss["loss_dict"] = mse(z_q_bar, z_e.detach())
        return (loss, x, x_tilde) if return_x else loss

    def _generator_loss_cl(self, x):
        return self._supervised_loss_cl(x)

    # ---- reference tensor-level API (vqvae.py:61-106) ----------------------------------------------
    def compute_supervised_loss(self, x, return_x=False):
        return self._supervised_loss_cl(self._as_cl(x), return_x)

    def decode(self, latents):
        """(N,num,h,w) int64 codes -> (N,C,H,W) in normalised space."""
        from .. import convstack
        self._require_gpu()
        y = self._decode_cl(latents.to(self.device))
        return convstack.cl_to_nchw(y, self.generator.out_channels)

    def configure_optimizers_and_checkpointers(self):
        o, c = super().configure_optimizers_and_checkpointers()
        if not self.use_codebook_ema:
            opt = build_optimizer(self.codebook, self.cfg, suffix="_G")
            o.append({"optimizer": opt, "scheduler": build_lr_scheduler(self.cfg, opt), "type": "generator"})
        os.makedirs(os.path.join(self.cfg.OUTPUT_DIR, "netC"), exist_ok=True)
        c.append({"checkpointer": Checkpointer(self.codebook, os.path.join(self.cfg.OUTPUT_DIR, "netC")),
                  "pretrained": self.cfg.MODEL.CODEBOOK.WEIGHTS})
        return o, c
