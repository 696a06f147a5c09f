#!/bin/bash
# Timeline of one captured step of any tool script (kernel, queue, start, duration, workgroups) + per-kernel time per step
# usage: bash tools/prof_step.sh <tag> <tool.py> [args...]   -> gpurun_out/<tag>_step_summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/step_$tag
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/step_$tag -o p -- python3 $R/"$@" > $R/gpurun_out/step_$tag.log 2>&1 || { tail -5 $R/gpurun_out/step_$tag.log; exit 1; }
python3 $R/tools/prof_summary.py $R/gpurun_out/step_$tag x y > $R/gpurun_out/${tag}_step_summary.txt 2>&1
rm -rf $R/gpurun_out/step_$tag
head -3 $R/gpurun_out/${tag}_step_summary.txt
