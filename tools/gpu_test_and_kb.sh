mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -5
python tools/kernel_bench.py > gpurun_out/kb.log 2>&1; grep -v amdgpu.ids gpurun_out/kb.log | tail -2
