# per-dispatch kernel trace of two eager training steps: durations of the hand-written GEMM / BatchNorm kernels by instantiation
# and grid (the stats summary hides the per-shape spread)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_step
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_step -- python3 $R/bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --entry step > $R/gpurun_out/trace_step.log 2>&1; echo rc=$?
cd $R
python - <<'PY'
import csv, glob, collections, re
f = glob.glob("/tmp/trace_step/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
pick = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    m = re.search(r"td::(conv1x1_mfma_kernel<[^>]*>|conv1x1_wgrad_kernel|bn_[a-z_]+kernel(<[^>]*>)?)", n)
    if m:
        pick[m.group(1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]), int(r.get("Grid_Size_Y", 1) or 1), int(r.get("Grid_Size_Z", 1) or 1)))
for k in sorted(pick):
    v = pick[k][len(pick[k]) // 2:]          # the last of the traced steps
    by = collections.defaultdict(list)
    for d, gx, gy, gz in v:
        by[(gx, gy, gz)].append(d)
    print("%s  calls/step %d  total %.1f us" % (k, len(v), sum(d for d, *_ in v) / 1e3))
    for (gx, gy, gz), ds in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print("   grid %8d x %4d x %2d  calls %3d  avg %7.2f us  total %8.1f us" % (gx, gy, gz, len(ds), sum(ds) / len(ds) / 1e3, sum(ds) / 1e3))
PY
