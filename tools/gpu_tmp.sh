#!/bin/bash
mkdir -p gpurun_out
echo base $(python tools/kernel_bench.py --only identity,fwd,bwd 2>&1 | tail -1)
python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; echo "tests: $(tail -1 gpurun_out/t.log)"
