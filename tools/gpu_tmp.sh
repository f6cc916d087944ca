#!/bin/bash
set -e
mkdir -p gpurun_out
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $2 > gpurun_out/$1.log 2>&1
  echo $1 $(tail -1 gpurun_out/$1.log | grep -o 'ms_per_step": [0-9.]*\|final_loss": [0-9.]*'); }
TD_NO_FUSED_BN=1 run nofuse_nofind "--miopen-find off"
run fuse_nofind "--miopen-find off"
TD_NO_FUSED_BN=1 run nofuse_find
run fuse_find
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
