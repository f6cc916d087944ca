# kernel timings + PMC passes for the photometric kernels (separate passes per the PMC slot rules)
mkdir -p gpurun_out
python tools/kernel_bench.py > gpurun_out/kb.log 2>&1; grep -v amdgpu.ids gpurun_out/kb.log | tail -2
python tools/kernel_bench.py --adversarial --only fwd,bwd > gpurun_out/kb_adv.log 2>&1; grep -v amdgpu.ids gpurun_out/kb_adv.log | tail -1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc/$tag -- python3 $R/tools/kernel_bench.py --iters 3 --only fwd,bwd > $R/gpurun_out/pmc_$tag.log 2>&1; echo "$tag rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc_summary.txt 2>&1; cat gpurun_out/pmc_summary.txt | tail -40
