"""Index samplers (reference: mono/datasets/loader/sampler.py).  The per-rank partition is the
data-parallel sharding of the training set: an epoch-seeded permutation, padded to a multiple of
samples_per_gpu * world, shuffled in units of samples_per_gpu, then rank r takes the contiguous
slice [r * n, (r + 1) * n)."""
import math

import numpy as np
import torch
from torch.distributed import get_rank, get_world_size
from torch.utils.data import DistributedSampler as _TorchDistributedSampler
from torch.utils.data import Sampler


def _ceil_to(n, unit):
    return int(math.ceil(n / unit)) * unit


class DistributedSampler(_TorchDistributedSampler):
    """Strided (rank::world) partition, optional epoch-seeded shuffle (reference :16-39)."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        super().__init__(dataset, num_replicas=num_replicas, rank=rank)
        self.shuffle = shuffle

    def __iter__(self):
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        order += order[: self.total_size - len(order)]
        assert len(order) == self.total_size
        mine = order[self.rank:self.total_size:self.num_replicas]
        assert len(mine) == self.num_samples
        return iter(mine)


class GroupSampler(Sampler):
    """Single-process sampler: shuffle within aspect-ratio groups (dataset.flag), pad each group
    to a multiple of samples_per_gpu, then shuffle whole batches (reference :42-79)."""

    def __init__(self, dataset, samples_per_gpu=1):
        assert hasattr(dataset, "flag")
        self.dataset = dataset
        self.samples_per_gpu = samples_per_gpu
        self.flag = dataset.flag.astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(_ceil_to(int(s), samples_per_gpu) for s in self.group_sizes)

    def __iter__(self):
        chunks = []
        for gid, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            members = np.where(self.flag == gid)[0]
            assert len(members) == size
            np.random.shuffle(members)
            pad = _ceil_to(int(size), self.samples_per_gpu) - len(members)
            chunks.append(np.concatenate([members, members[:pad]]))
        flat = np.concatenate(chunks)
        n_batches = len(flat) // self.samples_per_gpu
        batches = [flat[i * self.samples_per_gpu:(i + 1) * self.samples_per_gpu]
                   for i in np.random.permutation(range(n_batches))]
        out = torch.from_numpy(np.concatenate(batches)).long()
        assert len(out) == self.num_samples
        return iter(out)

    def __len__(self):
        return self.num_samples


class DistributedGroupSampler(Sampler):
    """Multi-process sampler used by _dist_train (reference :82-157)."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None):
        self.num_replicas = get_world_size() if num_replicas is None else num_replicas
        self.rank = get_rank() if rank is None else rank
        self.dataset = dataset
        self.samples_per_gpu = samples_per_gpu
        self.epoch = 0
        assert hasattr(dataset, "flag")
        self.flag = dataset.flag
        self.group_sizes = np.bincount(self.flag)
        unit = samples_per_gpu * self.num_replicas
        self.num_samples = sum(_ceil_to(int(s), unit) // self.num_replicas for s in self.group_sizes)
        self.total_size = self.num_samples * self.num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        unit = self.samples_per_gpu * self.num_replicas
        order = []
        for gid, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            members = np.where(self.flag == gid)[0]
            assert len(members) == size
            members = members[torch.randperm(int(size), generator=g).numpy()].tolist()
            members += members[: _ceil_to(int(size), unit) - len(members)]
            order += members
        assert len(order) == self.total_size
        spg = self.samples_per_gpu
        shuffled = []
        for b in torch.randperm(len(order) // spg, generator=g).tolist():
            shuffled.extend(order[b * spg:(b + 1) * spg])
        lo = self.num_samples * self.rank
        mine = shuffled[lo:lo + self.num_samples]
        assert len(mine) == self.num_samples
        return iter(mine)

    def __len__(self):
        return self.num_samples
