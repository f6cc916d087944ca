#!/bin/bash
set -e
mkdir -p gpurun_out
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $2 > gpurun_out/$1.log 2>&1
  echo $1 $(tail -1 gpurun_out/$1.log | grep -o 'ms_per_step": [0-9.]*\|final_loss": [0-9.]*'); }
run nofind "--miopen-find off"
run find
run find2
