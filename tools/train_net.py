#!/usr/bin/env python
"""Training / evaluation entry point with the reference's command line (tools/train_net.py:94-104,
engine/defaults.py:37-69):

    python tools/train_net.py --config-file configs/vqvae/PR-DVQVAE2.yaml --num-gpus 8 [--eval-only] [--resume] KEY VAL ...

Data: `--data-dir` points at a latent-code tree in the reference's on-disk format (for VideoTransformerModel) or
at a `.npy` of frames/clips (N,[T,]3,H,W) in [0,1] (for VQVAEModel); `--synthetic` uses random data of the
configured shape.  Dataset catalogs / image decoding of the reference are I/O and out of scope.
"""
import argparse
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import random

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from lvt_amd.config import get_cfg  # noqa: E402
from lvt_amd.data import DatasetMapper  # noqa: E402
from lvt_amd.data.latents import list_latent_videos, load_video_codes  # noqa: E402
from lvt_amd.data.samplers import TrainingSampler  # noqa: E402
from lvt_amd.engine.trainer import Trainer  # noqa: E402
from lvt_amd.evaluation import build_evaluator, inference_on_dataset  # noqa: E402
from lvt_amd.modeling import build_model  # noqa: E402
from lvt_amd.utils import comm  # noqa: E402


def default_argument_parser():
    p = argparse.ArgumentParser(description="lvt_amd training")
    p.add_argument("--config-file", default="", metavar="FILE")
    p.add_argument("--resume", action="store_true")
    p.add_argument("--eval-only", action="store_true")
    p.add_argument("--num-gpus", type=int, default=1)
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0)
    p.add_argument("--dist-url", default="tcp://127.0.0.1:{}".format(2 ** 15 + 2 ** 14 + hash(os.getuid()) % 2 ** 14))
    p.add_argument("--data-dir", default="")
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--max-iter", type=int, default=None)
    p.add_argument("--eval-batches", type=int, default=None,
                   help="--eval-only: truncate the test loader to this many batches per rank (default: the whole test set, as "
                        "the reference's test() does; synthetic data: 4)")
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p


def setup(args):
    cfg = get_cfg()
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    if cfg.MODEL.DEVICE == "cuda":
        cfg.MODEL.DEVICE = "cuda:%d" % comm.get_local_rank()
    cfg.freeze()
    os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
    if comm.is_main_process():
        with open(os.path.join(cfg.OUTPUT_DIR, "config.yaml"), "w") as f:
            f.write(cfg.dump())
    seed = cfg.SEED if cfg.SEED >= 0 else 0
    torch.manual_seed(seed + comm.get_rank())            # per-rank seed, defaults.py:113
    np.random.seed(seed + comm.get_rank())
    random.seed(seed + comm.get_rank())                  # DatasetMapper draws windows / slices with `random`
    return cfg


def data_iterator(cfg, args):
    """Infinite iterator of list[dict] batches, IMS_PER_BATCH / world per rank (data/build.py:62-74)."""
    world = comm.get_world_size()
    assert cfg.SOLVER.IMS_PER_BATCH % world == 0
    per_rank = cfg.SOLVER.IMS_PER_BATCH // world
    is_vt = cfg.MODEL.META_ARCHITECTURE == "VideoTransformerModel"
    seed = cfg.SEED if cfg.SEED >= 0 else 0
    if is_vt:
        mapper = DatasetMapper(cfg, True)
        if args.synthetic or not args.data_dir:
            v = cfg.MODEL.AUTOREGRESSIVE.VT
            rng = np.random.default_rng(seed)
            videos = [rng.integers(0, v.NV, (16, v.NC, 16, 16), dtype=np.int64) for _ in range(256)]
            load = lambda i: videos[i]                     # noqa: E731
            n = len(videos)
        else:
            vids = list_latent_videos(args.data_dir)
            load = lambda i: load_video_codes(vids[i][0], vids[i][1])      # noqa: E731
            n = len(vids)
        it = iter(TrainingSampler(n, seed=seed))
        while True:
            batch = []
            while len(batch) < per_rank:
                d = mapper({"image_sequence": load(next(it))})
                if d is not None:
                    batch.append(d)
            yield batch
    else:
        if args.synthetic or not args.data_dir:
            frames = np.random.default_rng(seed).random((1024, 3, 64, 64), dtype=np.float32)
        else:
            frames = np.load(args.data_dir, mmap_mode="r")
        key = "image_sequence" if frames.ndim == 5 else "image"
        it = iter(TrainingSampler(len(frames), seed=seed))
        while True:
            yield [{key: np.asarray(frames[next(it)], dtype=np.float32)} for _ in range(per_rank)]


def test_batches(cfg, args):
    """Finite test loader of --eval-only: the whole test set (or `--eval-batches` batches) in batches of IMS_PER_BATCH / world
    samples per rank, every rank a contiguous shard in order (the reference's InferenceSampler), unmapped for the transformer apart from the frame
    window (DatasetMapper(is_train=False): whole code clips, vidgen/data/dataset_mapper.py:113-149)."""
    world, rank = comm.get_world_size(), comm.get_rank()
    per_rank = max(1, cfg.SOLVER.IMS_PER_BATCH // world)
    seed = cfg.SEED if cfg.SEED >= 0 else 0
    if cfg.MODEL.META_ARCHITECTURE == "VideoTransformerModel":
        mapper = DatasetMapper(cfg, False)
        if args.synthetic or not args.data_dir:
            v = cfg.MODEL.AUTOREGRESSIVE.VT
            rng = np.random.default_rng(seed)
            videos = [rng.integers(0, v.NV, (16, v.NC, 16, 16), dtype=np.int64) for _ in range(world * per_rank * (args.eval_batches or 4))]
            load = lambda i: videos[i]                     # noqa: E731
            n = len(videos)
        else:
            vids = list_latent_videos(args.data_dir)
            load = lambda i: load_video_codes(vids[i][0], vids[i][1])      # noqa: E731
            n = len(vids)
        idx = _shard(n, rank, world, None if args.eval_batches is None else per_rank * args.eval_batches)
        for b in range(0, len(idx), per_rank):
            batch = [mapper({"image_sequence": load(i), "video_idx": i}) for i in idx[b:b + per_rank]]
            batch = [d for d in batch if d is not None]
            if batch:
                yield batch
    else:
        if args.synthetic or not args.data_dir:
            frames = np.random.default_rng(seed).random((world * per_rank * (args.eval_batches or 4), 3, 64, 64), dtype=np.float32)
        else:
            frames = np.load(args.data_dir, mmap_mode="r")
        key = "image_sequence" if frames.ndim == 5 else "image"
        idx = _shard(len(frames), rank, world, None if args.eval_batches is None else per_rank * args.eval_batches)
        for b in range(0, len(idx), per_rank):
            yield [{key: np.asarray(frames[i], dtype=np.float32), "video_idx": i} for i in idx[b:b + per_rank]]


def _shard(n, rank, world, limit):
    """Indices of this rank: a CONTIGUOUS shard, as the reference's InferenceSampler cuts the test set
    (vidgen/data/samplers/distributed_sampler.py: ceil(n / world) per rank), optionally truncated to `limit` samples."""
    per = -(-n // world)
    idx = list(range(min(rank * per, n), min((rank + 1) * per, n)))
    return idx if limit is None else idx[:limit]


def main(args):
    logging.basicConfig(level=logging.INFO if comm.is_main_process() else logging.WARNING,
                        format="[%(asctime)s] %(name)s %(levelname)s: %(message)s")
    cfg = setup(args)
    model = build_model(cfg)
    if args.eval_only:
        # the reference's `MyTrainer.test` (tools/train_net.py:35-57,76-86): load the checkpoints, run the test loader through
        # the model in inference mode and hand (inputs, outputs) to the evaluators cfg.TEST.EVALUATORS names
        _, checkpointers = model.configure_optimizers_and_checkpointers()
        for item in checkpointers:
            item["checkpointer"].resume_or_load(item["pretrained"], resume=False)
        evaluator = build_evaluator(cfg, cfg.DATASETS.TEST[0] if len(cfg.DATASETS.TEST) else "test")
        res = inference_on_dataset(model, test_batches(cfg, args), evaluator)
        if comm.is_main_process():
            logging.getLogger("lvt_amd").info("evaluation results: %s", dict(res))
        return res
    trainer = Trainer(cfg, model, data_iterator(cfg, args))
    trainer.resume_or_load(resume=args.resume)
    return trainer.train(args.max_iter)


def _worker(local_rank, args):
    os.environ["LOCAL_RANK"] = str(local_rank)
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", init_method=args.dist_url, world_size=args.num_gpus * args.num_machines,
                            rank=args.machine_rank * args.num_gpus + local_rank,
                            device_id=torch.device("cuda:%d" % local_rank))
    comm.synchronize()
    try:
        main(args)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    args = default_argument_parser().parse_args()
    print("Command Line Args:", args)
    if args.num_gpus * args.num_machines > 1:
        mp.spawn(_worker, nprocs=args.num_gpus, args=(args,))          # one process per GPU (launch.py:25-64)
    else:
        main(args)
