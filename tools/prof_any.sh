#!/bin/bash
# rocprofv3 kernel statistics of any tool script -> stdout (top kernels) and gpurun_out/<tag>_kernel_stats.csv
# usage: bash tools/prof_any.sh <tag> <tool.py> [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/"$@" > $R/gpurun_out/prof_$tag.log 2>&1 || { tail -5 $R/gpurun_out/prof_$tag.log; exit 1; }
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{r["Name"][:84]:84s} calls {int(r["Calls"]):5d} total {float(r["TotalDurationNs"])/1e6:8.2f} ms avg {float(r["AverageNs"])/1e3:8.1f} us {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
rm -rf $R/gpurun_out/prof_$tag
