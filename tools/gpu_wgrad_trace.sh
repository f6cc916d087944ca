# GPU time of the 1x1 weight-gradient kernels per shape, from a rocprofv3 kernel trace (the event-timed loop is launch-bound)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for t in default; do
  rm -rf /tmp/wg_$t
  rocprofv3 --kernel-trace --output-format csv -d /tmp/wg_$t -- python3 $R/tools/wgrad_probe.py --only-td > /dev/null 2>&1
  python3 - $t <<'PY'
import csv, glob, sys, collections
t = sys.argv[1]
f = glob.glob("/tmp/wg_%s/**/*kernel_trace.csv" % t, recursive=True)[0]
seq = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"])) for r in csv.DictReader(open(f))]
# 14 shapes x 23 calls (3 warm-up + 20 timed) of (wgrad, reduce), in order
w = [(d, g) for n, d, g in seq if "wgrad_kernel" in n]
r = [d for n, d, g in seq if "wgrad_reduce" in n]
tot = 0
for i in range(0, len(w), 23):
    ws = sorted(d for d, _ in w[i:i + 23]); rs = sorted(r[i:i + 23])
    wm, rm = ws[len(ws) // 2] / 1e3, rs[len(rs) // 2] / 1e3
    tot += wm + rm
    print("target %s  shape %2d  blocks %5d  wgrad %6.2f us  reduce %5.2f us" % (t, i // 23, w[i][1] // 256, wm, rm))
print("target %s  sum %.1f us" % (t, tot))
PY
done
