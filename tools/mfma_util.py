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
    ap.add_argument("--steady-steps", type=int, default=0, help="only the last N whole training steps (delimited by the identity-term kernel)")
    ap.add_argument("--json", default=None, help="write profiles/mfma.json (stamped with the sources and the workload; bench.py "
                                                 "quotes it only while both still match)")
    ap.add_argument("--steps", type=float, default=1.0, help="training steps covered by the selected dispatches")
    ap.add_argument("--workload", default="cfg_kitti_tripleD.py B=12 192x640")
    ap.add_argument("--source", default="")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.root, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    recs = []
    for f in files:
        recs += list(csv.DictReader(open(f)))
    if a.steady_steps:
        # whole steady-state steps: the identity-term kernel (photo_fwd_kernel<*, 0, *>) runs exactly once per training step
        marks = sorted({int(r["Dispatch_Id"]) for r in recs
                        if "photo_fwd_kernel" in (r.get("Kernel_Name") or "") and ", 0, " in (r.get("Kernel_Name") or "")})
        if len(marks) < a.steady_steps + 1:
            raise SystemExit("not enough steps in the counter file: %d identity kernels" % len(marks))
        lo, hi = marks[-a.steady_steps - 1], marks[-1]
        recs = [r for r in recs if lo <= int(r["Dispatch_Id"]) < hi]
        a.steps = float(a.steady_steps)
    elif a.last:
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
    own = [r for r in rows if "conv1x1_mfma_kernel" in r[4]]
    if own:
        oa = sum(r[0] for r in own)
        print("MfmaUtil over the hand-written td::conv1x1_mfma_kernel launches (time-weighted): %.1f %% (%.2f %% of the dispatch-busy cycles)"
              % (100 * sum(r[1] * r[0] for r in own) / oa, 100 * oa / tot_act))
    if a.json:
        import hashlib
        import json
        pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           "tripled-exploring-depth-estimation-with-self-supervised-representation-learning_amd")
        files = ["csrc/td_conv1x1.hip", "csrc/td_bn.hip", "hostside/mono/model/networks.py", "ops.py"]
        h = hashlib.sha256()
        for rel in files:
            with open(os.path.join(pkg, rel), "rb") as fh:
                h.update(fh.read())
        blob = {"source": a.source, "mfma_busy_cycles_per_step": raw["SQ_VALU_MFMA_BUSY_CYCLES"] / a.steps,
                "mfma_util_conv_kernels": round(conv_mfma / max(conv_act, 1), 4),
                "mfma_util_td_conv1x1": round(sum(r[1] * r[0] for r in own) / sum(r[0] for r in own), 4) if own else None,
                "clock_ghz": a.clock_ghz, "simds": a.cus * 4,
                "calibration": "8192^3 bf16 GEMM: counter 47.9 % vs wall-clock 44.2 % of 2.5 PFLOP/s (profiles/r02/pmc_mfma_gemm_calibration.txt)",
                "_stamp": {"sources_sha16": h.hexdigest()[:16], "sources": files, "workload": a.workload}}
        with open(a.json, "w") as fh:
            json.dump(blob, fh, indent=1)
    print("%8s %9s %7s  kernel" % ("share", "MfmaUtil", "calls"))
    for act, util, _, n, name in rows[:a.top]:
        print("%7.2f%% %8.1f%% %7d  %s" % (100 * act / tot_act, 100 * util, n, name[:120]))


if __name__ == "__main__":
    main()
