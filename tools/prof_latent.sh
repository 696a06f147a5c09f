#!/bin/bash
# kernel-time summary of the fused step at latent size $1 (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lat -o l -- python3 $R/tools/run_latents.py $1 > $R/gpurun_out/prof_lat.log 2>&1
cd $R && python3 tools/stats_top.py gpurun_out/prof_lat 2>&1 | head -${2:-30}
rm -rf gpurun_out/prof_lat
