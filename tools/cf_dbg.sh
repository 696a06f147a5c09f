cd /tmp && export TMPDIR=/tmp
for v in "" _d1 _d3; do
  export RBVAE_LIB=$GRAFT_REPO_ROOT/symbols-from-video_amd/librbvae_hip$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cf$v -o cf -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 100 > $GRAFT_REPO_ROOT/gpurun_out/prof_cf$v.log 2>&1
  echo "variant '$v': $(python3 $GRAFT_REPO_ROOT/tools/stats_top.py $GRAFT_REPO_ROOT/gpurun_out/prof_cf$v 40 | grep conv_first)"
done
