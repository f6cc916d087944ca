# rocprofv3 --pmc pass over the eager training step: MFMA-busy cycles per kernel (separate from any trace, per the PMC rules)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python3 $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --entry step > $R/gpurun_out/pmc_mfma.log 2>&1; echo rc=$?
cd $R
python tools/mfma_util.py /tmp/pmc_mfma --steady-steps 2 --json gpurun_out/mfma.json --source "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace over bench.py --no-graph, last two steady-state steps (tools/gpu_mfma_pmc.sh)" > gpurun_out/pmc_mfma_util_step.txt 2>&1
head -40 gpurun_out/pmc_mfma_util_step.txt | cut -c1-160
