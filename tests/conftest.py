import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# see tripled_amd/__init__.py: must be in the environment before the HIP runtime initialises
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """A per-test ceiling (pytest-timeout, where installed: it is in this image): the multi-process tests wait on children
    (mp.spawn / subprocess) and a rank that never arrives would otherwise hold the whole run until the driver's own limit.
    25 minutes is far above the slowest test (a bench.py child with its own 900 s limit)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(1500))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _one_stream_in_the_long_test_process():
    """The product forks the step's independent sub-networks onto side HIP streams (tripled_amd.streams, on by default).  In THIS
    process they stay on one stream unless a test turns the forks on itself (tests/test_hip_streams.py: eager forked-vs-serial
    parity; the forked step captured and replayed is tested through bench.py in a fresh process, the way the product runs it).
    Reason: replaying a forked graph that was captured ~300 tests into the pytest process segfaulted inside hipGraphLaunch
    (tests/test_hip_graph_step.py, first replay; the same test alone, bench.py and the train.py path replay their forked graphs
    without fault).  The cause is NOT established: a stream-aliasing defect found on the way is fixed and pinned
    (streams.side_stream; tests/test_streams_cpu.py, test_hip_streams.py), but the minimal negative control -- the pre-fix self-wait,
    captured and replayed -- does not fault either (tools/diag_self_wait_capture.py).  Until a full-suite run with the forks on has
    been done and read, this default stays -- DESIGN.md section 13."""
    if os.environ.get("TD_TEST_FORKS") == "1":      # the diagnostic run of tools/gpu_suite_forks_on.sh: forks stay on
        yield
        return
    try:
        import tripled_amd  # noqa: F401
        from tripled_amd import streams
    except Exception:      # noqa: BLE001 -- CPU-only collection without the package importable
        yield
        return
    old = streams.ENABLED
    streams.ENABLED = False
    yield
    streams.ENABLED = old
