"""Rank-strided infinite sampler (reference: vidgen/data/samplers/distributed_sampler.py:14-56).
All ranks walk the same seeded permutation stream and take indices[rank::world]: the data-parallel
sharding rule of the hot path (one shard of every global batch per GPU, no exchange of samples)."""
import itertools

import torch

from ..utils import comm


class TrainingSampler:
    def __init__(self, size, shuffle=True, seed=0, rank=None, world_size=None):
        assert size > 0
        self._size, self._shuffle, self._seed = size, shuffle, int(seed)
        self._rank = comm.get_rank() if rank is None else rank
        self._world = comm.get_world_size() if world_size is None else world_size

    def __iter__(self):
        yield from itertools.islice(self._infinite(), self._rank, None, self._world)

    def _infinite(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            if self._shuffle:
                yield from torch.randperm(self._size, generator=g).tolist()
            else:
                yield from range(self._size)
