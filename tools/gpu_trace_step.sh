# per-dispatch kernel trace of two eager training steps (for per-launch duration distributions; the stats summary hides them)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_step
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_step -- python3 $R/bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --entry step > $R/gpurun_out/trace_step.log 2>&1; echo rc=$?
cd $R
python - <<'PY'
import csv, glob, collections, json
f = glob.glob("/tmp/trace_step/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
pick = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    for key in ("bn_finalize_fwd", "bn_finalize_bwd", "bn_apply", "bn_dx", "bn_partials", "maxpool5", "conv1x1_wgrad_kernel", "conv1x1_wgrad_reduce"):
        if key in n:
            pick[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)), int(r.get("Grid_Size_Y", 1) or 1)))
out = {}
for k, v in pick.items():
    v = v[len(v) // 2:]          # the last of the traced steps
    by = collections.defaultdict(list)
    for d, gx, gy in v:
        by[(gx, gy)].append(d)
    out[k] = sorted(((sum(ds) / 1e3, len(ds), gx, gy, sum(ds) / len(ds) / 1e3) for (gx, gy), ds in by.items()), reverse=True)
    print(k, "calls", len(v), "total us %.1f" % (sum(d for d, _, _ in v) / 1e3))
    for tot, n, gx, gy, avg in out[k][:14]:
        print("   grid %7d x %4d  calls %3d  avg %.2f us  total %.1f us" % (gx, gy, n, avg, tot))
json.dump({k: v for k, v in out.items()}, open("gpurun_out/trace_step_bn.json", "w"))
PY
