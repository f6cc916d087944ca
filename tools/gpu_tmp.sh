#!/bin/bash
set -e
mkdir -p gpurun_out
export TD_DIST_BACKEND=gloo
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --miopen-find off --no-roofline > gpurun_out/n2_gloo.log 2>&1
tail -2 gpurun_out/n2_gloo.log | cut -c1-600
