# end-of-round evidence: bench line, rocprof kernel stats of the same command, calibrated PMC traffic of the hot kernels
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.log 2>&1; echo bench rc=$?
grep -v "amdgpu.ids\|Warning\|run_backward" gpurun_out/bench_full.log | tail -1 > gpurun_out/bench_line.json; cat gpurun_out/bench_line.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench /tmp/pmc_k /tmp/pmc_c
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1; echo prof rc=$?
f=$(ls /tmp/prof_bench/*/*kernel_stats.csv | head -1)
head -1 $f > $R/gpurun_out/bench_kernel_stats_td.csv; grep -E "(^\"|[^s])td::" $f >> $R/gpurun_out/bench_kernel_stats_td.csv
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_k/$pass -- python3 $R/tools/kernel_bench.py --iters 3 --only identity,fwd,bwd > $R/gpurun_out/pmc_$pass.log 2>&1; echo "$pass rc=$?"
  rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_c/$pass -- python3 $R/tools/kernel_bench.py --iters 3 --only calib > $R/gpurun_out/pmc_calib_$pass.log 2>&1; echo "calib $pass rc=$?"
done
cd $R
python tools/traffic_from_pmc.py /tmp/pmc_k /tmp/pmc_c > gpurun_out/traffic.json 2> gpurun_out/traffic.err; cat gpurun_out/traffic.json; tail -3 gpurun_out/traffic.err
python tools/pmc_summary.py /tmp/pmc_k > gpurun_out/pmc_summary.txt 2>&1
cat gpurun_out/bench_kernel_stats_td.csv | cut -c1-200
