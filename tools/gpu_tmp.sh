#!/bin/bash
set -e
mkdir -p gpurun_out
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --miopen-find off --split-timing $2 > gpurun_out/$1.log 2>&1
  echo $1 $(grep "split timing" gpurun_out/$1.log) $(tail -1 gpurun_out/$1.log | grep -o 'ms_per_step": [0-9.]*'); }
run noflat --no-flat
TD_FLAT_ALIGN=256 run flat_align256
TD_FLAT_LOWP=0 run flat_fp32only
TD_FLAT_LOWP=0 TD_FLAT_ALIGN=256 run flat_fp32only_align256
