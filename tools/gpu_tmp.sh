mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.log 2>&1; echo rc=$?
grep -v "amdgpu.ids\|Warning\|run_backward" gpurun_out/bench_full.log | tail -1
