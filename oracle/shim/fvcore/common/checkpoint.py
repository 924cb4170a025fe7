import os
import torch


class Checkpointer:
    def __init__(self, model, save_dir="", **checkpointables):
        self.model = model
        self.save_dir = save_dir

    def save(self, name, **kwargs):
        data = {"model": self.model.state_dict()}
        data.update(kwargs)
        torch.save(data, os.path.join(self.save_dir, name + ".pth"))

    def resume_or_load(self, path, resume=True):
        if path:
            ckpt = torch.load(path, map_location="cpu")
            self.model.load_state_dict(ckpt["model"])
            return ckpt
        return {}


class PeriodicCheckpointer:
    def __init__(self, checkpointer, period, max_iter=None):
        self.checkpointer, self.period, self.max_iter = checkpointer, int(period), max_iter

    def step(self, iteration, **kwargs):
        iteration = int(iteration)
        if (iteration + 1) % self.period == 0:
            self.checkpointer.save("model_{:07d}".format(iteration), **kwargs)
        if self.max_iter is not None and iteration >= self.max_iter - 1:
            self.checkpointer.save("model_final", **kwargs)
