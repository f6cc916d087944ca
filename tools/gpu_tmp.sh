timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -4
python tools/kernel_bench.py --only identity,fwd,bwd 2>&1 | grep -v amdgpu.ids | tail -1
