"""tools/trace_overlap.py on a synthetic kernel trace: two chains that overlap, a gap, the step cut at the identity kernel."""
import csv
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("trace_overlap", os.path.join(ROOT, "tools", "trace_overlap.py"))
trace_overlap = importlib.util.module_from_spec(spec)
spec.loader.exec_module(trace_overlap)

MARK = "void td::photo_fwd_kernel<2, 0, true>(float const*)"


def _write(path, rows):
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kind", "Queue_Id", "Stream_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for q, name, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", q, 0, name, s, e])


def _step(t0):
    """One step of 1000 ns: main chain 0-300 (mark) + 300-600, side chain 100-500 overlapping it, gap 600-800, tail 800-1000."""
    return [(1, MARK, t0, t0 + 300), (2, "side_a", t0 + 100, t0 + 500), (1, "main_b", t0 + 300, t0 + 600),
            (1, "tail", t0 + 800, t0 + 1000)]


def test_concurrency_profile_of_a_synthetic_trace(tmp_path):
    rows = []
    for k in range(5):                       # 4 complete steps + the mark of a fifth
        rows += _step(10_000 + 1000 * k)
    path = tmp_path / "kernel_trace.csv"
    _write(path, rows)
    sel, t_end = trace_overlap.cut_steps(trace_overlap.read_trace(str(path)), 3)
    assert len(sel) == 12 and t_end == 14_000
    p = trace_overlap.profile(sel, t_end)
    assert p["span_ns"] == 3000 and p["launches"] == 12
    assert p["sum_ns"] == 3 * (300 + 400 + 300 + 200)
    assert p["busy_ns"] == 3 * 800 and p["idle_ns"] == 3 * 200
    assert p["overlap_ns"] == 3 * 400
    assert p["depth_ns"] == {0: 600, 1: 1200, 2: 1200}
    assert p["queues"]["1/0"] == {"kernel_ns": 3 * 800, "launches": 9}
    assert p["queues"]["2/0"] == {"kernel_ns": 3 * 400, "launches": 3}
    text = trace_overlap.report(p, 3)
    assert "33.3 % of sum" in text and "k = 2" in text


def test_too_short_a_trace_is_an_error(tmp_path):
    path = tmp_path / "kernel_trace.csv"
    _write(path, _step(0))
    import pytest
    with pytest.raises(ValueError, match="not enough steps"):
        trace_overlap.cut_steps(trace_overlap.read_trace(str(path)), 3)


def test_cli_writes_the_json(tmp_path, capsys):
    rows = []
    for k in range(3):
        rows += _step(1000 * k)
    path = tmp_path / "kernel_trace.csv"
    _write(path, rows)
    out = tmp_path / "o.json"
    trace_overlap.main([str(path), "2", "--json", str(out)])
    import json
    blob = json.load(open(out))
    assert blob["steps"] == 2 and blob["busy_ns"] == 1600
    assert "busy" in capsys.readouterr().out
