mkdir -p gpurun_out
for a in 0 1 2; do echo "ablate=$a"; TD_ABLATE=$a python tools/kernel_bench.py --only identity,fwd 2>&1 | grep -v amdgpu.ids | tail -1; done
