R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_sq/p$i -- python3 $R/tools/kernel_bench.py --iters 3 --only identity,fwd,bwd > $R/gpurun_out/pmc_sq_$i.log 2>&1; echo "pass $i rc=$?"
done
cd $R
python tools/pmc_summary.py /tmp/pmc_sq > gpurun_out/pmc_sq_summary.txt 2>&1
grep -A17 "td::photo_bwd_kernel<2>" gpurun_out/pmc_sq_summary.txt | grep "WAIT\|ACTIVE_INST_ANY\|WAVE_CYCLES\|INSTS_VALU"
