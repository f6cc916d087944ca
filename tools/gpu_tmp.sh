#!/bin/bash
set -e
mkdir -p gpurun_out
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/b.log 2>&1
tail -1 gpurun_out/b.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --h2d > gpurun_out/b_h2d.log 2>&1
tail -1 gpurun_out/b_h2d.log | grep -o 'ms_per_step": [0-9.]*'
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
