"""Dispatch book-keeping (tripled_amd/dispatch.py): the GPU tests and bench.py rely on it to prove that a step ran on the
hand-written kernels (`fallbacks == 0`, strict mode raises)."""
import logging

import pytest

import tripled_amd  # noqa: F401
from tripled_amd import dispatch


@pytest.fixture(autouse=True)
def _clean():
    dispatch.reset()
    prev = dispatch.set_strict(False)
    yield
    dispatch.set_strict(prev)
    dispatch.reset()


def test_counters_and_snapshot():
    dispatch.hip("td_bn_fwd")
    dispatch.hip("td_bn_fwd")
    dispatch.hip("td_photo_fwd", 4)
    assert dispatch.hip_calls["td_bn_fwd"] == 2 and dispatch.hip_calls["td_photo_fwd"] == 4
    snap = dispatch.snapshot()
    assert snap == {"hip_calls": {"td_bn_fwd": 2, "td_photo_fwd": 4}, "fallbacks": {}}
    dispatch.reset()
    assert not dispatch.hip_calls and not dispatch.fallbacks


def test_fallback_is_logged_once_per_site_and_counted(caplog):
    with caplog.at_level(logging.WARNING, logger="tripled_amd"):
        for _ in range(3):
            dispatch.fallback("Conv3x3.pad", "7 channels")
        dispatch.fallback("BatchNorm", "")
    assert dispatch.fallbacks == {"Conv3x3.pad": 3, "BatchNorm": 1}
    lines = [r.getMessage() for r in caplog.records]
    assert len(lines) == 2 and "Conv3x3.pad" in lines[0] and "7 channels" in lines[0]


def test_strict_mode_raises_and_restores():
    with dispatch.strict():
        with pytest.raises(dispatch.FallbackError, match="maxpool5.*odd layout"):
            dispatch.fallback("maxpool5", "odd layout")
        with dispatch.strict(False):
            dispatch.fallback("maxpool5", "odd layout")      # inner scope relaxed
        with pytest.raises(dispatch.FallbackError):
            dispatch.fallback("maxpool5")
    dispatch.fallback("maxpool5")                            # outside: not strict again
    assert dispatch.fallbacks["maxpool5"] == 4               # a raising fallback is still counted


def test_native_check_counts_the_entry_point():
    from tripled_amd import native
    native.check(0, "td_smooth_fwd")
    assert dispatch.hip_calls["td_smooth_fwd"] == 1
    with pytest.raises(RuntimeError):
        native.check(-1, "td_smooth_fwd")
