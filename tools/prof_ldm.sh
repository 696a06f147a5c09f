cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ldm -o l -- python3 $GRAFT_REPO_ROOT/tools/run_ldm.py > $GRAFT_REPO_ROOT/gpurun_out/prof_ldm.log 2>&1
cd $GRAFT_REPO_ROOT && tail -3 gpurun_out/prof_ldm.log && python tools/stats_top.py gpurun_out/prof_ldm 12
