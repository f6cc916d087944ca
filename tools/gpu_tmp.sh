mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -2
python tools/kernel_bench.py --only identity,fwd,bwd 2>&1 | grep -v amdgpu.ids | tail -1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --miopen-find 2>&1 | grep -v "amdgpu.ids\|Warning\|run_backward" | tail -2
