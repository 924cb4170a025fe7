"""Minimal checkpointer with the on-disk convention the reference gets from fvcore
(`{"model": state_dict}` in `<save_dir>/<name>.pth`; model_{iter:07d}.pth / model_final.pth;
`last_checkpoint` marker).  The reference constructs one per sub-network (ae.py:231-238)."""
import logging
import os

import torch


class Checkpointer:
    def __init__(self, model, save_dir="", save_to_disk=True, **checkpointables):
        if hasattr(model, "module") and isinstance(model, torch.nn.parallel.DistributedDataParallel):
            model = model.module
        self.model = model
        self.save_dir = save_dir
        self.save_to_disk = save_to_disk
        self.checkpointables = dict(checkpointables)

    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return
        data = {"model": self.model.state_dict()}
        for k, obj in self.checkpointables.items():
            data[k] = obj.state_dict()
        data.update(kwargs)
        os.makedirs(self.save_dir, exist_ok=True)
        basename = "{}.pth".format(name)
        torch.save(data, os.path.join(self.save_dir, basename))
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(basename)

    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
                return os.path.join(self.save_dir, f.read().strip())
        except IOError:
            return ""

    def load(self, path):
        if not path:
            return {}
        if not os.path.isfile(path):
            raise FileNotFoundError("Checkpoint {} not found!".format(path))
        ckpt = torch.load(path, map_location="cpu")
        state = ckpt.pop("model")
        state = {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}
        incompatible = self.model.load_state_dict(state, strict=False)
        self.last_incompatible = incompatible           # (missing_keys, unexpected_keys), also logged like fvcore does
        if incompatible.missing_keys:
            logging.getLogger("lvt_amd").warning("checkpoint %s: keys missing from the file: %s", path,
                                                 ", ".join(incompatible.missing_keys))
        if incompatible.unexpected_keys:
            logging.getLogger("lvt_amd").warning("checkpoint %s: keys not used by the model: %s", path,
                                                 ", ".join(incompatible.unexpected_keys))
        for k, obj in self.checkpointables.items():
            if k in ckpt:
                obj.load_state_dict(ckpt.pop(k))
        return ckpt

    def resume_or_load(self, path, resume=True):
        if resume and self.has_checkpoint():
            path = self.get_checkpoint_file()
        return self.load(path)


class PeriodicCheckpointer:
    def __init__(self, checkpointer, period, max_iter=None):
        self.checkpointer, self.period, self.max_iter = checkpointer, int(period), max_iter

    def step(self, iteration, **kwargs):
        iteration = int(iteration)
        if (iteration + 1) % self.period == 0:
            self.checkpointer.save("model_{:07d}".format(iteration), iteration=iteration, **kwargs)
        if self.max_iter is not None and iteration >= self.max_iter - 1:
            self.checkpointer.save("model_final", iteration=iteration, **kwargs)
