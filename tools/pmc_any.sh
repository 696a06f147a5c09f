#!/bin/bash
# SQ / LDS / TCC counter passes over any tool script; run on the GPU box.
# usage: bash tools/pmc_any.sh <kernel-name-filter> <tool.py> [args...]   -> per-kernel counter averages + derived figures
R=${GRAFT_REPO_ROOT:-/root/repo}
filt=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmca$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmca$i -- python3 $R/"$@" > $R/gpurun_out/pmca$i.log 2>&1 || { tail -5 $R/gpurun_out/pmca$i.log; exit 1; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmca$i "$filt"
  rm -rf $R/gpurun_out/pmca$i
done
