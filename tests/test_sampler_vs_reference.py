"""The data-parallel sharding of the training set against the REAL reference samplers (mono/datasets/loader/sampler.py:15-157,
loaded from /root/reference as a stand-alone module: it needs only torch and numpy).  Same dataset flags, world size, rank, epoch
and numpy seed -> the same index stream, element for element.  Skipped where the reference checkout is absent (the GPU box)."""
import importlib.util
import os

import numpy as np
import pytest

import tripled_amd  # noqa: F401
from mono.datasets.loader import sampler as mine

REF_FILE = "/root/reference/mono/datasets/loader/sampler.py"
pytestmark = pytest.mark.skipif(not os.path.isfile(REF_FILE), reason="reference checkout not present")


def _reference():
    spec = importlib.util.spec_from_file_location("_reference_sampler", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    import sys
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    return mod


class _DS:
    def __init__(self, flags):
        self.flag = np.asarray(flags, dtype=np.int64)

    def __len__(self):
        return len(self.flag)


def _flags(n, groups, seed):
    return np.random.RandomState(seed).randint(0, groups, size=n)


@pytest.mark.parametrize("n,groups,spg,world", [(50, 1, 4, 3), (697, 1, 12, 8), (101, 2, 4, 2), (64, 3, 8, 4), (100, 1, 12, 8),
                                                (39810, 1, 12, 8)])
def test_distributed_group_sampler_streams_are_the_reference_streams(n, groups, spg, world):
    ref = _reference()
    ds = _DS(_flags(n, groups, seed=n))
    for rank in range(world):
        a, b = ref.DistributedGroupSampler(ds, spg, world, rank), mine.DistributedGroupSampler(ds, spg, world, rank)
        assert len(a) == len(b)
        for epoch in (0, 1, 7):
            a.set_epoch(epoch)
            b.set_epoch(epoch)
            ia, ib = [int(i) for i in a], [int(i) for i in b]
            assert ia == ib, (rank, epoch)


@pytest.mark.parametrize("n,world,shuffle", [(50, 3, True), (50, 3, False), (697, 8, False), (13, 4, True)])
def test_distributed_sampler_streams_are_the_reference_streams(n, world, shuffle):
    ref = _reference()
    ds = _DS(np.zeros(n))
    for rank in range(world):
        a = ref.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle)
        b = mine.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle)
        assert len(a) == len(b)
        for epoch in (0, 3):
            a.set_epoch(epoch)
            b.set_epoch(epoch)
            assert [int(i) for i in a] == [int(i) for i in b], (rank, epoch)


@pytest.mark.parametrize("n,groups,spg", [(50, 1, 4), (101, 2, 4), (64, 3, 12), (30, 1, 12)])
def test_group_sampler_streams_are_the_reference_streams(n, groups, spg):
    """GroupSampler draws from numpy's global generator (sampler.py:57-70): with the same seed in front of each iteration the
    two give the same stream."""
    ref = _reference()
    ds = _DS(_flags(n, groups, seed=3 * n))
    a, b = ref.GroupSampler(ds, spg), mine.GroupSampler(ds, spg)
    assert len(a) == len(b)
    for seed in (0, 11):
        np.random.seed(seed)
        ia = [int(i) for i in a]
        np.random.seed(seed)
        ib = [int(i) for i in b]
        assert ia == ib, seed


def test_a_set_smaller_than_its_pad_fails_like_the_reference():
    """7 samples, 12 per GPU, 8 ranks: the pad (89) is longer than the set, `indice[:extra]` cannot supply it and the reference's
    length assertion fires (sampler.py:142) -- the mirror keeps that error behaviour instead of inventing a cyclic pad."""
    ref = _reference()
    ds = _DS(np.zeros(7))
    for mod in (ref, mine):
        s = mod.DistributedGroupSampler(ds, 12, 8, 0)
        with pytest.raises(AssertionError):
            list(s)
