"""Process-group helpers with the reference's names (vidgen/utils/comm.py:21-79).  One process per
GPU; the tensor process group is RCCL ("nccl" backend on ROCm) over xGMI, gloo on CPU test runs."""
import torch.distributed as dist


def _ready():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if _ready() else 1


def get_rank():
    return dist.get_rank() if _ready() else 0


def get_local_rank():
    import os
    return int(os.environ.get("LOCAL_RANK", 0)) if _ready() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    if _ready() and dist.get_world_size() > 1:
        dist.barrier()
