#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of the default bench under the environment given on the command line:
#   bash tools/pmc_traffic_env.sh NAME VAR=VALUE ...   -> gpurun_out/pmc_NAME/{pmc_traffic.json,pmc_traffic_per_kernel.txt}
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/pmc_$name; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --no-cpu --no-roofline --steps 40 > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --no-cpu --no-roofline --steps 40 > $O/write.log 2>&1 || exit 1
cd $R && python3 tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json $O/pmc_traffic_per_kernel.txt > /dev/null
rm -rf $O/fetch $O/write
grep "wgrad_gemm_k" $O/pmc_traffic_per_kernel.txt
