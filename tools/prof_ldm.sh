#!/bin/bash
# rocprofv3 kernel statistics of the frozen LDM encoder at 512x512 frames -> gpurun_out/prof_ldm_stats.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ldm
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ldm -o l -- python3 $R/tools/run_ldm.py > $R/gpurun_out/prof_ldm.log 2>&1 || { tail -5 $R/gpurun_out/prof_ldm.log; exit 1; }
f=$(find $R/gpurun_out/prof_ldm -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/prof_ldm_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d}  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
rm -rf $R/gpurun_out/prof_ldm
