cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cf -o cf -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 200 > $GRAFT_REPO_ROOT/gpurun_out/prof_cf.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/stats_top.py gpurun_out/prof_cf 2>&1 40
