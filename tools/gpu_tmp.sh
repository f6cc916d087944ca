timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --miopen-find off 2>&1 | grep -v "amdgpu.ids\|Warning\|run_backward" | tail -1 | cut -c1-330
