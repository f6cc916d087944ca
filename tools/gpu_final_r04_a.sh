# round-4 final evidence, part A: MFMA PMC pass at HEAD -> profiles/mfma.json (stamped), then the headline bench line (so that it
# carries the PMC block and the calibrated traffic), then rocprof kernel stats of the same bench command
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
bash tools/gpu_mfma_pmc.sh > gpurun_out/r04_mfma_pmc.txt 2>&1
cp gpurun_out/mfma.json profiles/mfma.json
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.log 2>&1; echo bench rc=$?
grep -v "amdgpu.ids\|Warning\|run_backward" gpurun_out/bench_full.log | tail -1 > gpurun_out/r04_bench_line_final.json; cut -c1-200 gpurun_out/r04_bench_line_final.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1; echo prof rc=$?
f=$(ls /tmp/prof_bench/*/*kernel_stats.csv | head -1)
head -1 $f > $R/gpurun_out/r04_bench_kernel_stats_td_final.csv; grep -E "(^\"|[^s])td::" $f >> $R/gpurun_out/r04_bench_kernel_stats_td_final.csv
cp $f $R/gpurun_out/r04_bench_kernel_stats_all_final.csv
cd $R
head -12 gpurun_out/pmc_mfma_util_step.txt | cut -c1-160
