# rocprofv3 kernel trace of the eager training step; steady-state steps are cut out of the trace.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_step
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_step -- python3 $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_step.log 2>&1; echo rc=$?
cd $R
f=$(ls /tmp/prof_step/*/*kernel_trace.csv | head -1)
python tools/kstats_from_trace.py $f 3 gpurun_out/step_kernel_stats.csv
python tools/kstats_groups.py gpurun_out/step_kernel_stats.csv 3
grep -v "amdgpu.ids\|Warning\|run_backward" gpurun_out/prof_step.log | tail -1 | cut -c1-300
