#!/bin/bash
# SQ counter passes over the isolated conv2-shaped weight-gradient GEMM (tools/abl_wgrad.py 16 7); GPU box, from /tmp.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcw$i -- python3 $R/tools/abl_wgrad.py 16 7 > $R/gpurun_out/pmcw$i.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcw$i wgrad_gemm
  rm -rf $R/gpurun_out/pmcw$i
done
