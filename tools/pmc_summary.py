#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if "td::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        grid = r.get("Grid_Size", "")
        acc[(short, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), ctrs in sorted(acc.items()):
    print(k, "grid", grid)
    for c, v in sorted(ctrs.items()):
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
