#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_bn.py -x -q > gpurun_out/t_bn.log 2>&1; echo "bn tests: $(tail -1 gpurun_out/t_bn.log)"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/b.log 2>&1; echo batched $(tail -1 gpurun_out/b.log | grep -o 'ms_per_step": [0-9.]*\|final_loss": [0-9.]*')
python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; echo "tests: $(tail -1 gpurun_out/t.log)"
