"""Minimal training loop with the reference's step semantics (vidgen/engine/trainer.py:56-128,
train_loop.py:112-133, defaults.py:273-310), minus its per-step host synchronisations:

  * `run_step`: next batch -> `model(data, mode='supervised')` -> sum of the loss dict -> backward ->
    every ACCUMULATION_STEPS: all optimizers step, then all zero_grad; schedulers step once per iteration.
  * losses are kept on the device; they are fetched (one D2H copy) only every `log_period` iterations, where the
    finiteness check of `_detect_anomaly` is also applied.  The reference does `.item()` + a gloo gather of a
    pickled dict every step, which serialises the ranks.
  * checkpoints: one Checkpointer per sub-network, `model_{iter:07d}.pth` / `model_final.pth`, rank 0 only.
"""
import logging
import time

import torch

from ..utils import comm
from ..utils.checkpoint import PeriodicCheckpointer
from ..utils.events import EventStorage


class Trainer:
    def __init__(self, cfg, model, data_iter, log_period=20):
        self.cfg, self.model, self.data_iter, self.log_period = cfg, model, data_iter, log_period
        self.optimizers, self.checkpointers = model.configure_optimizers_and_checkpointers()
        if comm.get_world_size() > 1:
            model.wrap_parallel(device_ids=[comm.get_local_rank()], broadcast_buffers=False)
        model.train()
        self.start_iter, self.max_iter = 0, cfg.SOLVER.MAX_ITER
        self.accumulation_steps = cfg.SOLVER.ACCUMULATION_STEPS
        self.periodic = [PeriodicCheckpointer(c["checkpointer"], cfg.SOLVER.CHECKPOINT_PERIOD, self.max_iter)
                         for c in self.checkpointers] if comm.is_main_process() else []
        self.logger = logging.getLogger("lvt_amd")
        self.iter = 0

    def resume_or_load(self, resume=True):
        for item in self.checkpointers:
            item["checkpointer"].resume_or_load(item["pretrained"], resume=resume)

    def run_step(self):
        assert self.model.training, "model was changed to eval mode!"
        data = next(self.data_iter)
        loss_dict = self.model(data, mode="supervised")
        losses = sum(loss_dict.values())
        losses.backward()
        if (self.iter + 1) % self.accumulation_steps == 0:
            for item in self.optimizers:
                item["optimizer"].step()
            for item in self.optimizers:
                item["optimizer"].zero_grad()
        for item in self.optimizers:
            item["scheduler"].step()
        return loss_dict

    def train(self, max_iter=None):
        max_iter = self.max_iter if max_iter is None else max_iter
        t0, last = time.perf_counter(), None
        with EventStorage(self.start_iter) as storage:
            for self.iter in range(self.start_iter, max_iter):
                last = self.run_step()
                if (self.iter + 1) % self.log_period == 0 or self.iter == max_iter - 1:
                    vals = {k: float(v.detach()) for k, v in last.items()}          # the only D2H sync
                    if not all(torch.isfinite(torch.tensor(list(vals.values())))):
                        raise FloatingPointError("Loss became infinite or NaN at iteration={}! {}".format(self.iter, vals))
                    storage.put_scalars(**vals)
                    if comm.is_main_process():
                        dt = (time.perf_counter() - t0) / (self.iter + 1 - self.start_iter)
                        self.logger.info("iter %d  %s  %.4f s/it", self.iter,
                                         "  ".join("%s: %.5f" % kv for kv in vals.items()), dt)
                for pc in self.periodic:
                    pc.step(self.iter)
                storage.step()
        return last
