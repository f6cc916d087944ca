# round-4 final evidence, part B: SQ wait / issue / active counters of the photometric kernels (kernel_bench) and of every
# hand-written kernel of the eager step; bench lines of the other configs
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq_k /tmp/pmc_sq_s
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_sq_k -- python3 $R/tools/kernel_bench.py --iters 3 --only identity,fwd,bwd > $R/gpurun_out/pmc_sq_k.log 2>&1; echo "sq kernels rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_sq_s -- python3 $R/bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --entry step > $R/gpurun_out/pmc_sq_s.log 2>&1; echo "sq step rc=$?"
cd $R
python tools/pmc_summary.py /tmp/pmc_sq_k > gpurun_out/r04_pmc_sq_photometric.txt 2>&1; tail -30 gpurun_out/r04_pmc_sq_photometric.txt | cut -c1-200
python tools/pmc_wait_summary.py /tmp/pmc_sq_s > gpurun_out/r04_pmc_wait_shares_step.txt 2>&1; head -34 gpurun_out/r04_pmc_wait_shares_step.txt
for c in cfg_kitti_fm cfg_kitti_tripleD_320x1024 cfg_kitti_fm_joint_inpaint_disentangle_distill_full_colorize; do
  python bench.py --config config/$c.py --no-cpu-baseline > gpurun_out/bench_$c.log 2>&1; echo "$c rc=$?"
  grep -v "amdgpu.ids\|Warning\|run_backward" gpurun_out/bench_$c.log | tail -1 > gpurun_out/r04_bench_line_$c.json; cut -c1-160 gpurun_out/r04_bench_line_$c.json
done
