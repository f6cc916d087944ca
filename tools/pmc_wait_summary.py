#!/usr/bin/env python3
"""Per-kernel share of wave cycles spent parked at s_waitcnt / stalled at issue / issuing, from a rocprofv3 --pmc pass
with SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES (counter_collection.csv).  Hand-written kernels only."""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "td::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            calls[short] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
print("%-60s %7s %8s %8s %8s" % ("kernel", "calls", "wait", "istall", "active"))
for name, c in rows[:30]:
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-60s %7d %7.0f%% %7.0f%% %7.0f%%" % (name[:60], calls[name], 100 * c.get("SQ_WAIT_ANY", 0) / wc,
                                               100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc))
