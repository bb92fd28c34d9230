"""``LocalShuffleSampler`` -- mirror of the reference's ``nerf/local_shuffler.py``: every model of the model-averaging regime draws
only from ITS OWN part of the dataset (a DistributedSampler would shuffle across parts).  Host-side index bookkeeping; no device work.
"""
import random
from typing import Iterable, List, Optional, Union

import numpy as np
import torch.distributed as dist
from torch.utils.data import Dataset, Sampler


class LocalShuffleSampler(Sampler):
    """indices: either the number of parts (the dataset is cut into that many contiguous, equally long parts, the remainder going to the
    last one) or one part id per sample.  Every epoch yields a seeded shuffle (seed + epoch) of the rank's own part, truncated to the
    size of the smallest part unless ``allow_imbalance`` (local_shuffler.py:19-92; with ``shuffle=False`` the reference yields the whole
    index range, which is kept)."""

    def __init__(self, dataset: Dataset, indices: Union[List[int], int] = 4, rank: Optional[int] = None, shuffle: bool = True, seed: int = 0,
                 allow_imbalance=False) -> None:
        if rank is None:
            if not dist.is_available() or not dist.is_initialized():
                raise RuntimeError("Requires distributed package to be available")
            rank = dist.get_rank()
        if not isinstance(indices, Iterable):
            length, num_replicas = len(dataset), int(indices)
            part = length // num_replicas
            indices = np.minimum(np.arange(length) // max(part, 1), num_replicas - 1).astype(np.int32)
        else:
            indices = list(indices)
            num_replicas = max(indices) + 1
        if rank >= num_replicas or rank < 0:
            raise ValueError("Invalid rank {}, rank should be in the interval [0, {}]".format(rank, num_replicas - 1))
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.epoch, self.drop_last = 0, False
        self.samples = [[] for _ in range(num_replicas)]
        for i, part_id in enumerate(indices):
            self.samples[int(part_id)].append(i)
        self.min_sample = None if allow_imbalance else min(len(s) for s in self.samples)
        self.shuffle, self.seed = shuffle, seed

    def __iter__(self):
        if not self.shuffle:
            return iter(range(len(self.dataset)))
        own = list(self.samples[self.rank])
        random.seed(self.seed + self.epoch)
        random.shuffle(own)
        return iter(own[: self.min_sample])

    def __len__(self):
        return self.min_sample if self.min_sample is not None else len(self.samples[self.rank])

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch
