"""Evaluation callers of the hot path (reference: vidgen/evaluation/evaluator.py:14-166,
codes_extractor.py:36-53, mse_evaluation.py:28-47, bits_evaluation.py:28-58).  They only consume the
dicts `model(inputs)` returns in inference mode; statistics are accumulated on the device and reduced across
ranks with one all-reduce at the end (the reference gathers pickled python objects over gloo)."""
import logging
import math
import os
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from ..data.latents import save_video_codes
from ..layers import all_reduce_sum_
from ..utils import comm


class DatasetEvaluator:
    def reset(self):
        pass

    def process(self, inputs, outputs):
        pass

    def evaluate(self):
        pass


class DatasetEvaluators(DatasetEvaluator):
    def __init__(self, evaluators):
        self._evaluators = list(evaluators)

    def reset(self):
        for e in self._evaluators:
            e.reset()

    def process(self, inputs, outputs):
        for e in self._evaluators:
            e.process(inputs, outputs)

    def evaluate(self):
        results = OrderedDict()
        for e in self._evaluators:
            r = e.evaluate()
            if comm.is_main_process() and r is not None:
                for k, v in r.items():
                    assert k not in results, "Different evaluators produce results with the same key {}".format(k)
                    results[k] = v
        return results


class CodesExtractor(DatasetEvaluator):
    """Writes `output['latent']` (T, nc, h, w) as one .npy per frame under
    <output_dir>/<dataset_name>/video_<idx>/<frame>.npy -- the training data format of the transformer."""

    def __init__(self, dataset_name, distributed=True, output_dir=None):
        self._root = os.path.join(output_dir, dataset_name)

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            latent = out["latent"]
            if latent.dim() == 3:
                latent = latent.unsqueeze(1)
            save_video_codes(self._root, int(inp["video_idx"]), latent.detach().cpu().numpy())

    def evaluate(self):
        comm.synchronize()
        return {}


class _SumEvaluator(DatasetEvaluator):
    def __init__(self, dataset_name, distributed=True, output_dir=None):
        self._logger = logging.getLogger(__name__)
        self._distributed = distributed
        self.reset()

    def reset(self):
        self._acc = None                      # [sum, count] on the device of the first output

    def _add(self, s, n):
        v = torch.stack([s.double().reshape(()), torch.tensor(float(n), dtype=torch.float64, device=s.device)])
        self._acc = v if self._acc is None else self._acc + v

    def _totals(self):
        if self._acc is None:                 # a rank that saw no sample still takes part in the all-reduce
            dist_on = self._distributed and torch.distributed.is_available() and torch.distributed.is_initialized()
            dev = "cuda" if (dist_on and torch.cuda.is_available() and torch.distributed.get_backend() == "nccl") else "cpu"
            self._acc = torch.zeros(2, dtype=torch.float64, device=dev)
        acc = self._acc.clone()
        if self._distributed:
            all_reduce_sum_(acc)
        return float(acc[0]), float(acc[1])


class MSEEvaluator(_SumEvaluator):
    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            rec = out["reconstruction"].detach()
            tgt = torch.as_tensor(inp["image"] if "image" in inp else inp["image_sequence"], device=rec.device)
            if rec.is_cuda:       # sum of squared differences on the device: lvt_mse_fwd (F.mse_loss(reduction="sum"))
                from ..hip import ew
                sq = ew.mse_fwd(rec.contiguous().float(), tgt.to(torch.float32).contiguous(), denom=1.0)
            else:
                sq = F.mse_loss(rec, tgt.to(rec.dtype), reduction="sum")
            self._add(sq, tgt.numel())

    def evaluate(self):
        s, n = self._totals()
        if not comm.is_main_process():
            return None
        results = OrderedDict({"reconstruction": {"MSE": s / n}})
        self._logger.info(results)
        return results


class BitsEvaluator(_SumEvaluator):
    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            logits = out["logits"]                                   # nc, nv, T, H, W
            target = torch.as_tensor(inp["image_sequence"], device=logits.device).transpose(0, 1)   # nc, T, H, W
            keep = ~out["ignore_mask"].expand(target.size(0), -1, -1, -1)
            if logits.is_cuda:
                # lvt_xent_fwd on token-major rows (vidgen/evaluation/bits_evaluation.py:28-58 sums F.cross_entropy over the
                # kept positions): (nc, nv, P) -> (nc * P, nv) by lvt_permute3, ignored positions carry the ignore index
                from ..hip import tx
                nc, nv = logits.shape[:2]
                P = target[0].numel()
                rows = tx.permute3(logits.contiguous().float().view(nc, nv, P), (nv * P, 1, P), (nc, P, nv)).view(nc * P, nv)
                tgt = torch.where(keep, target, torch.full_like(target, -100)).reshape(1, nc * P).contiguous()
                # the SUM over the kept rows (bits_evaluation.py:36-40), from the per-row terms in fp64: `mean * count` would be
                # 0/0 = NaN for a sample whose positions are all ignored, and rounds the mean to fp32 first
                _, _, _, row_loss = tx.xent_fwd(rows, tgt[0], 0, 1, nc * P, -100, 1.0, want_rows=True)
                self._add(row_loss.double().sum(), int(keep.sum()))
            else:
                ce = F.cross_entropy(logits.permute(1, 0, 2, 3, 4).unsqueeze(0), target.unsqueeze(0), reduction="none")[0]
                self._add(ce[keep].sum(), int(keep.sum()))

    def evaluate(self):
        s, n = self._totals()
        if not comm.is_main_process():
            return None
        results = OrderedDict({"likelihood": {"bits_per_dim": (s / math.log(2)) / n}})
        self._logger.info(results)
        return results


class VTSampler(DatasetEvaluator):
    """Decodes and saves the videos `VideoTransformerModel` sampled in inference mode (reference:
    vidgen/evaluation/vt_sampler.py:18-81): builds the VQ-VAE named by cfg.TEST.VT_SAMPLER.VQ_VAE.*, and for every input writes
    `<output_dir>/samples/<dataset>/video_<sample>_<video_idx>/{codes.npy, <frame>.png}`.  The decode runs on the HIP path
    (`vqvae.decode` -> gather + ResDecoder); an empty weight path keeps the initialised weights, as fvcore's Checkpointer does."""

    def __init__(self, cfg, dataset_name, distributed=True, output_dir=None, vqvae=None):
        from ..config import get_cfg
        from ..modeling import build_model
        from ..utils.checkpoint import Checkpointer
        self._logger = logging.getLogger(__name__)
        self._dataset_name, self._distributed, self._output_dir = dataset_name, distributed, output_dir
        vs = cfg.TEST.VT_SAMPLER.VQ_VAE
        vq_cfg = get_cfg()
        vq_cfg.merge_from_file(vs.CFG)
        vq_cfg.MODEL.DEVICE = cfg.MODEL.DEVICE
        vq_cfg.OUTPUT_DIR = cfg.OUTPUT_DIR
        self.vqvae = vqvae if vqvae is not None else build_model(vq_cfg)
        for module, path in ((self.vqvae.encoder, vs.ENCODER_WEIGHTS), (self.vqvae.generator, vs.GENERATOR_WEIGHTS),
                             (self.vqvae.codebook, vs.CODEBOOK_WEIGHTS)):
            if path:
                Checkpointer(module).resume_or_load(path, resume=False)
        self.vqvae.set_generator_requires_grad(False)
        self.vqvae.eval()
        self.scale_to_zeroone = vq_cfg.INPUT.SCALE_TO_ZEROONE

    @torch.no_grad()
    def process(self, inputs, outputs):
        from PIL import Image
        for inp, out in zip(inputs, outputs):
            v_idx = inp["video_idx"]
            for sample_idx, sample in enumerate(out["samples"]):        # each (nc, T, h, w), or (1, T, h, w) -> (T, h, w)
                sample = sample.squeeze(0)
                if sample.dim() == 4:
                    sample = sample.transpose(0, 1).contiguous()         # (T, nc, h, w)
                code = sample.detach().cpu().numpy()
                frames = self.vqvae.back_normalizer(self.vqvae.decode(sample))           # (T, 3, H, W)
                if self.scale_to_zeroone:
                    frames = frames * 255
                frames = frames.clamp_(0.0, 255.0).permute(0, 2, 3, 1).contiguous().cpu().numpy().astype(np.uint8)
                video_dir = os.path.join(self._output_dir, "samples", self._dataset_name, "video_%d_%s" % (sample_idx, v_idx))
                os.makedirs(video_dir, exist_ok=True)
                np.save(os.path.join(video_dir, "codes.npy"), code)
                for frame_idx in range(len(frames)):
                    path = os.path.join(video_dir, "%d.png" % frame_idx)
                    for attempt in range(10):                            # (vt_sampler.py:74-81: shared file systems hiccup)
                        try:
                            Image.fromarray(frames[frame_idx]).save(path)
                            break
                        except OSError:
                            if attempt == 9:
                                raise
                            time.sleep(3)

    def evaluate(self):
        if self._distributed:
            comm.synchronize()
        return None


def build_evaluator(cfg, dataset_name, output_folder=None):
    """Substring dispatch on cfg.TEST.EVALUATORS like tools/train_net.py:35-57 of the reference."""
    if output_folder is None:
        output_folder = os.path.join(cfg.OUTPUT_DIR, "inference")
    evs = []
    if "CodesExtractor" in cfg.TEST.EVALUATORS:
        evs.append(CodesExtractor(dataset_name, True, output_folder))
    if "MSEEvaluator" in cfg.TEST.EVALUATORS:
        evs.append(MSEEvaluator(dataset_name, True, output_folder))
    if "BitsEvaluator" in cfg.TEST.EVALUATORS:
        evs.append(BitsEvaluator(dataset_name, True, output_folder))
    if "VTSampler" in cfg.TEST.EVALUATORS:
        evs.append(VTSampler(cfg, dataset_name, True, output_folder))
    if not evs:
        raise NotImplementedError(cfg.TEST.EVALUATORS)
    return evs[0] if len(evs) == 1 else DatasetEvaluators(evs)


def inference_on_dataset(model, data_loader, evaluator):
    """Run `model(inputs)` in eval mode / no_grad over an iterable of batches and evaluate."""
    logger = logging.getLogger(__name__)
    evaluator = evaluator or DatasetEvaluators([])
    evaluator.reset()
    was_training = model.training
    model.eval()
    n, t0 = 0, time.perf_counter()
    with torch.no_grad():
        for inputs in data_loader:
            outputs = model(inputs)
            evaluator.process(inputs, outputs)
            n += len(inputs)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    model.train(was_training)
    logger.info("Total inference time: %.3f s (%.6f s / sample per device, on %d devices)",
                time.perf_counter() - t0, (time.perf_counter() - t0) / max(n, 1), comm.get_world_size())
    results = evaluator.evaluate()
    return {} if results is None else results
