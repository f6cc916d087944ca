#!/usr/bin/env python3
"""Steady-state per-step kernel statistics from a rocprofv3 kernel_trace CSV.  Steps are delimited by
the identity-term kernel (td::photo_fwd_kernel<*, 0, *>), which runs exactly once per training step;
only the last N complete steps are counted (skips warm-up and MIOpen find-mode tuning)."""
import collections
import csv
import sys

path, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
out = sys.argv[3] if len(sys.argv) > 3 else None
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "photo_fwd_kernel" in r[2] and ", 0, " in r[2]]
if len(marks) < nsteps + 1:
    sys.exit("not enough steps in trace: %d identity kernels" % len(marks))
lo, hi = marks[-nsteps - 1], marks[-1]
sel = rows[lo:hi]
wall = (rows[hi][0] - rows[lo][0]) / 1e6 / nsteps
acc = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    acc[n][0] += e - s
    acc[n][1] += 1
tot = sum(v[0] for v in acc.values())
print("steps %d  wall %.2f ms/step  kernel-busy %.2f ms/step  launches/step %d" % (nsteps, wall, tot / 1e6 / nsteps, len(sel) // nsteps))
if out:
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for n, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            w.writerow([n, c, t, t / c, 100.0 * t / tot])
