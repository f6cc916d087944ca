# PMC passes for the photometric kernels (separate passes per the PMC slot rules)
mkdir -p gpurun_out
ONLY=${ONLY:-bwd}
python tools/kernel_bench.py --only $ONLY 2>&1 | grep -v amdgpu.ids | tail -1
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_WAVE32_VALU SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc/$tag -- python3 $R/tools/kernel_bench.py --iters 3 --only $ONLY > $R/gpurun_out/pmc_$tag.log 2>&1; echo "$tag rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc_summary.txt 2>&1; cat gpurun_out/pmc_summary.txt | tail -70
