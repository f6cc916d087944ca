#!/usr/bin/env python3
"""How much of a training step's kernel time overlaps: concurrency profile of a rocprofv3 kernel_trace CSV.

The forked step (tripled_amd.streams: three sub-networks as parallel branches of the HIP graph) does not make any kernel
faster, it lets kernels of different chains run side by side.  The per-kernel statistics (tools/kstats_from_trace.py) cannot
show that; this tool reads the per-dispatch start / end timestamps and reports, per steady-state step:

  span          first kernel start -> next step's first kernel start (what ms_per_step sees)
  busy          length of the UNION of the kernels' intervals (the device had at least one kernel in flight)
  sum           sum of the kernels' own durations (what a one-stream step needs at the least)
  overlap       sum - busy: kernel time hidden behind another kernel
  idle          span - busy: nothing in flight (launch gaps, dependency stalls)
  at depth k    time with exactly k kernels in flight
  per queue     busy time and launches of every Queue_Id / Stream_Id the trace names (graph branches land on several)

Measured caveat (profiles/r04/overlap_forks_on_v1.txt): rocprofv3's kernel tracing serialises the dispatches of a graph's parallel
branches on this stack (one kernel in flight for 94 % of the span, the traced step takes 45 ms against 27 ms untraced), so under the
tracer busy ~ sum and the overlap itself has to be read from ms_per_step with the forks on / off; what the trace does give is the
split of the kernel time over the branches (per queue).

Steps are delimited like tools/kstats_from_trace.py does: by the identity-term kernel (td::photo_fwd_kernel<*, 0, *>), which
runs exactly once per step; only the last N complete steps are counted.

  python tools/trace_overlap.py <kernel_trace.csv> [N=3] [--json out.json]
"""
import collections
import csv
import json
import sys


def is_step_mark(name):
    return "photo_fwd_kernel" in name and ", 0, " in name


def read_trace(path):
    """[(start_ns, end_ns, kernel name, queue key)] sorted by start."""
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            q = "/".join(str(r[k]) for k in ("Queue_Id", "Stream_Id") if k in r and r[k] != "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], q or "-"))
    rows.sort()
    return rows


def cut_steps(rows, nsteps):
    """The dispatches of the last ``nsteps`` complete steps and the start of the step after them."""
    marks = [i for i, r in enumerate(rows) if is_step_mark(r[2])]
    if len(marks) < nsteps + 1:
        raise ValueError("not enough steps in the trace: %d identity kernels, %d needed" % (len(marks), nsteps + 1))
    lo, hi = marks[-nsteps - 1], marks[-1]
    # the identity kernel is not a step's first kernel, but every step is cut at the same phase, which is all the per-step
    # averages need
    return rows[lo:hi], rows[hi][0]


def profile(sel, t_end):
    """Concurrency profile of the dispatches ``sel`` (sorted by start) up to ``t_end``."""
    t0 = sel[0][0]
    events = []
    for s, e, _, _ in sel:
        e = min(e, t_end)
        if e > s:
            events.append((s, 1))
            events.append((e, -1))
    events.sort()
    depth_ns = collections.Counter()
    depth, last = 0, t0
    for t, d in events:
        if t > last:
            depth_ns[depth] += t - last
            last = t
        depth += d
    if t_end > last:
        depth_ns[0] += t_end - last
    span = t_end - t0
    busy = span - depth_ns.get(0, 0)
    total = sum(min(e, t_end) - s for s, e, _, _ in sel if min(e, t_end) > s)
    queues = collections.defaultdict(lambda: [0, 0])
    for s, e, _, q in sel:
        queues[q][0] += max(0, min(e, t_end) - s)
        queues[q][1] += 1
    return {"span_ns": span, "busy_ns": busy, "sum_ns": total, "overlap_ns": total - busy, "idle_ns": span - busy,
            "depth_ns": dict(sorted(depth_ns.items())), "launches": len(sel),
            "queues": {q: {"kernel_ns": v[0], "launches": v[1]} for q, v in sorted(queues.items(), key=lambda kv: -kv[1][0])}}


def report(p, nsteps):
    ms = lambda ns: ns / 1e6 / nsteps
    lines = ["steps %d   launches/step %d" % (nsteps, p["launches"] // nsteps),
             "span    %8.3f ms/step" % ms(p["span_ns"]),
             "busy    %8.3f ms/step   (union of the kernel intervals)" % ms(p["busy_ns"]),
             "sum     %8.3f ms/step   (kernels' own durations)" % ms(p["sum_ns"]),
             "overlap %8.3f ms/step   (sum - busy: hidden behind another kernel, %.1f %% of sum)"
             % (ms(p["overlap_ns"]), 100.0 * p["overlap_ns"] / max(p["sum_ns"], 1)),
             "idle    %8.3f ms/step   (span - busy: nothing in flight)" % ms(p["idle_ns"]),
             "time with k kernels in flight:"]
    for k, ns in p["depth_ns"].items():
        lines.append("   k = %-2d %8.3f ms/step  %5.1f %%" % (k, ms(ns), 100.0 * ns / max(p["span_ns"], 1)))
    lines.append("per queue / stream (kernel time, launches per step):")
    for q, v in p["queues"].items():
        lines.append("   %-12s %8.3f ms/step  %5d" % (q, ms(v["kernel_ns"]), v["launches"] // nsteps))
    return "\n".join(lines)


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    out = argv[argv.index("--json") + 1] if "--json" in argv else None
    if out in args:
        args.remove(out)
    path, nsteps = args[0], int(args[1]) if len(args) > 1 else 3
    sel, t_end = cut_steps(read_trace(path), nsteps)
    p = profile(sel, t_end)
    print(report(p, nsteps))
    if out:
        with open(out, "w") as fh:
            json.dump({"steps": nsteps, **p}, fh, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
