# rocprofv3 kernel stats (all kernels) of the graph-replayed bench: 5 timed + 2 warm-up steps
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --entry step > $R/gpurun_out/prof_bench.log 2>&1; echo prof rc=$?
f=$(ls /tmp/prof_bench/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/bench_kernel_stats_all.csv
cd $R
n=$(python -c "
import csv,sys
print(sum(int(r['Calls']) for r in csv.DictReader(open('gpurun_out/bench_kernel_stats_all.csv')) if 'photo_fwd_kernel<2, 0' in r['Name']))")
echo "steps in the trace (identity launches): $n"
python tools/kstats_groups.py gpurun_out/bench_kernel_stats_all.csv $n > gpurun_out/bench_kernel_groups.txt 2>&1; head -30 gpurun_out/bench_kernel_groups.txt
