#!/bin/bash
set -u; mkdir -p gpurun_out/pmc; export PYTHONUNBUFFERED=1
R=$PWD; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc/p$i -o pass -- python $R/tools/pmc_workload.py > $R/gpurun_out/pmc/p$i.log 2>&1; echo "pass $i ($grp) rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_summary.tsv $(ls gpurun_out/pmc/p*/pass_results.db) 2>&1 | tail -20
