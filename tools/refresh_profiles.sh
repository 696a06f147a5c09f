#!/bin/bash
# Round artefacts on ONE box: default bench line, rocprofv3 --stats of the same command, FETCH/WRITE PMC passes.
# usage (GPU box): bash tools/refresh_profiles.sh    -> files under gpurun_out/refresh/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh; mkdir -p $O
cd $R && python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
tail -1 $O/bench_default.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-cpu > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --no-cpu --steps 40 > $O/fetch.log 2>&1 || exit 1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --no-cpu --steps 40 > $O/write.log 2>&1 || exit 1
echo write done
cd $R && python3 tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json $O/pmc_traffic_per_kernel.txt && python3 tools/prof_summary.py $O/stats > $O/step_summary.txt 2>&1
rm -f $O/*/*kernel_trace.csv $O/fetch/*counter_collection.csv.bak
ls -la $O
