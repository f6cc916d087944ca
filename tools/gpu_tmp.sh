timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -2
python tools/kernel_bench.py 2>&1 | grep -v amdgpu.ids | tail -1
