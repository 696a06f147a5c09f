#!/bin/bash
# SQ counter passes over tools/abl_gemm.py (gather GEMM, conv2-forward shape); run on the GPU box from /tmp.
# usage: bash tools/pmc_gemm.sh <tool.py> [args...]   -> prints per-kernel counter averages
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcg$i -- python3 $R/"$@" > $R/gpurun_out/pmcg$i.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcg$i gemm
done
