"""tripled_amd.streams on the host: the stream bookkeeping of Branch with a stand-in for torch.cuda.Stream that behaves like
the runtime's pool (32 streams handed out round-robin, so two Stream() objects can BE the same HIP stream).  The GPU side of
the same property is tests/test_hip_streams.py::test_a_branch_never_lands_on_the_current_stream."""
import itertools

import pytest
import torch

tripled_amd = pytest.importorskip("tripled_amd")
from tripled_amd import streams  # noqa: E402


class _PoolStream:
    """32 handles per device, round-robin, like c10's stream pool; handle 0 is the default stream and never pooled."""
    _next = itertools.count()
    waits = []

    def __init__(self, device=None, handle=None):
        self.device = device
        self.cuda_stream = handle if handle is not None else 0x1000 + (next(_PoolStream._next) % 32) * 0x10

    def wait_stream(self, other):
        _PoolStream.waits.append((self.cuda_stream, other.cuda_stream))


@pytest.fixture()
def pool(monkeypatch):
    current = {"s": _PoolStream(handle=0)}
    monkeypatch.setattr(streams.torch.cuda, "Stream", _PoolStream)
    monkeypatch.setattr(streams.torch.cuda, "current_stream", lambda *a: current["s"])
    monkeypatch.setattr(streams.torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(streams, "_side", {})
    _PoolStream._next = itertools.count()
    _PoolStream.waits = []
    return current


def test_side_streams_are_cached_distinct_and_never_the_default(pool):
    dev = torch.device("cuda", 0)
    s0, s1, s2 = (streams.side_stream(dev, i) for i in range(3))
    assert len({s0.cuda_stream, s1.cuda_stream, s2.cuda_stream, 0}) == 4
    assert streams.side_stream(dev, 0) is s0 and streams.side_stream(dev, 1) is s1 and streams.side_stream(dev, 2) is s2
    # a device given without an index resolves to the current device and hits the same cache entries
    assert streams.side_stream(torch.device("cuda"), 0) is s0


def test_a_branch_moves_off_a_current_stream_that_aliases_its_cached_stream(pool):
    dev = torch.device("cuda", 0)
    b0 = streams.Branch(dev, 0)
    b1 = streams.Branch(dev, 1)
    first0, first1 = b0.stream.cuda_stream, b1.stream.cuda_stream
    # the caller's stream cycles through the whole pool (a capture stream created many Stream() calls later)
    for k in range(70):
        pool["s"] = _PoolStream()
        n0, n1 = streams.Branch(dev, 0), streams.Branch(dev, 1)
        ids = {pool["s"].cuda_stream, n0.stream.cuda_stream, n1.stream.cuda_stream}
        assert len(ids) == 3 and 0 not in ids, (k, ids)
    # and the aliasing case was really among them
    pool["s"] = _PoolStream(handle=streams.side_stream(dev, 0).cuda_stream)
    moved = streams.Branch(dev, 0)
    assert moved.stream.cuda_stream != pool["s"].cuda_stream
    assert moved.stream.cuda_stream != streams.side_stream(dev, 1).cuda_stream
    assert first0 != first1


def test_join_from_the_fork_stream_does_not_wait_on_itself(pool):
    dev = torch.device("cuda", 0)
    b = streams.Branch(dev, 0)
    pool["s"] = _PoolStream(handle=0x7770)
    _PoolStream.waits = []
    b.join({"x": [torch.zeros(2)]})                    # CPU tensors are skipped by record_stream
    assert _PoolStream.waits == [(0x7770, b.stream.cuda_stream)]
    # a join issued while the branch's stream is current (nested use) records no self-dependency
    pool["s"] = _PoolStream(handle=b.stream.cuda_stream)
    _PoolStream.waits = []
    b.join(())
    assert _PoolStream.waits == []


def test_no_free_stream_is_an_error_not_a_spin(pool, monkeypatch):
    dev = torch.device("cuda", 0)

    class _OneStream(_PoolStream):
        def __init__(self, device=None, handle=None):
            super().__init__(device, handle=0x2000 if handle is None else handle)

    monkeypatch.setattr(streams.torch.cuda, "Stream", _OneStream)
    pool["s"] = _OneStream()
    with pytest.raises(RuntimeError, match="no free HIP stream"):
        streams.Branch(dev, 0)


def test_disabled_or_host_tensors_do_not_fork(monkeypatch):
    monkeypatch.setattr(streams, "ENABLED", True)
    assert streams.enabled(torch.zeros(1)) is False
    monkeypatch.setattr(streams, "ENABLED", False)
    assert not streams.enabled(torch.zeros(1))
