#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc run (counter_collection.csv).

  python tools/mfma_util.py <dir-with-*counter_collection.csv> [--cus 256]

MfmaUtil(kernel) = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE * CUs * 4 SIMDs)   (the gfx94x formula of
rocprof's derived_counters.xml; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over all SIMDs, GRBM_GUI_ACTIVE counts
the cycles the dispatch kept the GPU busy -- MI355X_MICROARCH.md, 'rocprofv3 PMC slots').
"""
import argparse
import collections
import csv
import glob
import os
import re

CONV = r"igemm|Cijk|gridwise|naive_conv|[Ww]inograd|kernel_gemm|grouped_conv|batched_gemm|conv"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--cus", type=int, default=256)
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--last", type=int, default=0, help="only the last N dispatches (steady-state steps after MIOpen's find)")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.root, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    recs = []
    for f in files:
        recs += list(csv.DictReader(open(f)))
    if a.last:
        ids = sorted({int(r["Dispatch_Id"]) for r in recs})
        keep = set(ids[-a.last:])
        recs = [r for r in recs if int(r["Dispatch_Id"]) in keep]
    seen = set()
    for r in recs:
        name = r.get("Kernel_Name") or r.get("Kernel Name")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            calls[name] += 1
        if "Start_Timestamp" in r and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            acc[name]["__ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = []
    for name, c in acc.items():
        act = c.get("GRBM_GUI_ACTIVE", 0.0)
        if act <= 0:
            continue
        ns = c.get("__ns", 0.0)
        if ns > 0:      # wall duration of the dispatch x nominal clock, instead of the GRBM cycle count
            util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (ns * a.clock_ghz * a.cus * 4)
        else:
            util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act * a.cus * 4)
        rows.append((act, util, c.get("SQ_BUSY_CU_CYCLES", 0.0), calls[name], name))
    rows.sort(reverse=True)
    tot_act = sum(r[0] for r in rows)
    conv = [r for r in rows if re.search(CONV, r[4])]
    conv_act = sum(r[0] for r in conv)
    conv_mfma = sum(r[1] * r[0] for r in conv)
    print("dispatch-busy cycles: total %.3e, convolution/GEMM kernels %.3e (%.1f %%)" % (tot_act, conv_act, 100 * conv_act / tot_act))
    print("MfmaUtil over the convolution/GEMM kernels (time-weighted): %.1f %%" % (100 * conv_mfma / max(conv_act, 1)))
    print("MfmaUtil over ALL kernels of the step (time-weighted):     %.1f %%" % (100 * sum(r[1] * r[0] for r in rows) / tot_act))
    raw = {k: sum(c.get(k, 0.0) for c in acc.values()) for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CU_CYCLES", "__ns")}
    print("raw sums:", raw)
    print("%8s %9s %7s  kernel" % ("share", "MfmaUtil", "calls"))
    for act, util, _, n, name in rows[:a.top]:
        print("%7.2f%% %8.1f%% %7d  %s" % (100 * act / tot_act, 100 * util, n, name[:120]))


if __name__ == "__main__":
    main()
