"""Data-parallel sharding of the training set (reference: mono/datasets/loader/sampler.py:82-157):
rank r of R takes the r-th contiguous slice of an epoch-seeded permutation, padded to a multiple of samples_per_gpu * R."""
import numpy as np

import tripled_amd  # noqa: F401
from mono.datasets.loader.sampler import DistributedGroupSampler, DistributedSampler, GroupSampler


class _DS:
    def __init__(self, n, flags=None):
        self.flag = np.zeros(n, dtype=np.int64) if flags is None else np.asarray(flags, dtype=np.int64)

    def __len__(self):
        return len(self.flag)


def test_ranks_partition_the_epoch():
    n, spg, world = 50, 4, 3
    ds = _DS(n)
    shards = []
    for r in range(world):
        s = DistributedGroupSampler(ds, spg, world, r)
        s.set_epoch(5)
        idx = list(s)
        assert len(idx) == len(s) == 20 and len(idx) % spg == 0      # ceil(50 / 12) * 12 / 3
        shards.append(idx)
    seen = [i for sh in shards for i in sh]
    assert set(seen) == set(range(n))                                # every sample is visited
    counts = np.bincount(seen, minlength=n)
    assert counts.max() <= 2 and counts.sum() == 60                  # the pad repeats at most once
    # ranks are disjoint apart from the pad duplicates
    assert sum(len(set(a) & set(b)) for i, a in enumerate(shards) for b in shards[i + 1:]) <= 60 - n


def test_epoch_seed_and_determinism():
    ds = _DS(32)
    a = DistributedGroupSampler(ds, 2, 2, 0)
    b = DistributedGroupSampler(ds, 2, 2, 0)
    a.set_epoch(1)
    b.set_epoch(1)
    assert list(a) == list(b)
    b.set_epoch(2)
    assert list(a) != list(b)


def test_groups_stay_together_within_a_batch():
    flags = [0] * 10 + [1] * 7
    ds = _DS(len(flags), flags)
    s = DistributedGroupSampler(ds, 2, 2, 1)
    idx = list(s)
    for i in range(0, len(idx), 2):
        assert flags[idx[i]] == flags[idx[i + 1]]
    g = list(GroupSampler(ds, 2))
    assert len(g) == 10 + 8
    for i in range(0, len(g), 2):
        assert flags[int(g[i])] == flags[int(g[i + 1])]


def test_strided_sampler_covers_the_set():
    ds = _DS(11)
    got = []
    for r in range(4):
        s = DistributedSampler(ds, num_replicas=4, rank=r, shuffle=False)
        got += list(s)
    assert set(got) == set(range(11)) and len(got) == 12
