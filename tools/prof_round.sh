#!/bin/bash
# Kernel statistics + HBM-side traffic of the other configurations (run on the GPU box): cfg 3, native 4x88x160, LDM encoder,
# and the PMC view of the two halo kernels.  Output: gpurun_out/r03/*.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03; mkdir -p $O
cd $R
for cfg in "cfg3 tools/run_cfg3.py bf16" "native tools/run_native.py bf16" "ldm tools/run_ldm.py"; do
  set -- $cfg; tag=$1; shift
  bash tools/prof_any.sh $tag "$@" > $O/${tag}_top_kernels.txt 2>&1 || exit 1
  cp gpurun_out/${tag}_kernel_stats.csv $O/${tag}_kernel_stats.csv
done
bash tools/pmc_any.sh conv_halo tools/time_conv_halo.py 0,2,4 > $O/conv_halo_pmc.txt 2>&1 || exit 1
bash tools/pmc_any.sh deconv_halo tools/time_deconv_halo.py 2,4 > $O/deconv_halo_pmc.txt 2>&1 || exit 1
python3 tools/time_conv_halo.py > $O/conv_halo_times.txt 2>&1
python3 tools/time_deconv_halo.py > $O/deconv_halo_times.txt 2>&1
python3 tools/run_ldm.py > $O/ldm_run.txt 2>&1
ls $O
