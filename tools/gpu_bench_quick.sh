mkdir -p gpurun_out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_graph.log 2>&1; echo rc=$?
grep -v amdgpu.ids gpurun_out/b_graph.log | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_eager -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_eager.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_eager | head -20
