# calibrated PMC traffic of the photometric kernels (FETCH_SIZE / WRITE_SIZE in separate passes + the dword-stream calibration)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_k /tmp/pmc_c
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_k/$pass -- python3 $R/tools/kernel_bench.py --iters 3 --only identity,fwd,bwd > $R/gpurun_out/pmc_$pass.log 2>&1; echo "$pass rc=$?"
  rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_c/$pass -- python3 $R/tools/kernel_bench.py --iters 3 --only calib > $R/gpurun_out/pmc_calib_$pass.log 2>&1; echo "calib $pass rc=$?"
done
cd $R
python tools/traffic_from_pmc.py /tmp/pmc_k /tmp/pmc_c > gpurun_out/traffic.json 2> gpurun_out/traffic.err; tail -8 gpurun_out/traffic.json; tail -3 gpurun_out/traffic.err
python tools/pmc_summary.py /tmp/pmc_k > gpurun_out/pmc_summary.txt 2>&1
