#!/bin/bash
# Round artefacts on ONE box: default bench line, rocprofv3 kernel-trace stats of its headline legs (--no-cpu --no-others: the
# other configurations launch the same kernel templates at other shapes and would mix into the per-kernel averages), FETCH/WRITE PMC
# passes, SQ (matrix-core busy / occupancy) PMC pass.   usage (GPU box): bash tools/refresh_profiles.sh
# -> files under gpurun_out/refresh/ ; copy the summaries into profiles/ (named per round).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh; mkdir -p $O
cd $R && python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
tail -1 $O/bench_default.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-cpu --no-others > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --no-cpu --no-roofline --no-others --steps 40 > $O/fetch.log 2>&1 || exit 1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --no-cpu --no-roofline --no-others --steps 40 > $O/write.log 2>&1 || exit 1
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o q -- python3 $R/bench.py --no-cpu --no-roofline --no-others --steps 40 > $O/sq.log 2>&1 || exit 1
echo sq done
python3 - > $O/pmc_meta.json <<PY
import json, socket, subprocess, time
head = open("$R/.git/HEAD").read().strip() if __import__("os").path.exists("$R/.git/HEAD") else "snapshot of the working tree (no .git on the GPU box)"
print(json.dumps({"taken": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()), "host": socket.gethostname(), "tree": head,
                  "command": "tools/refresh_profiles.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* passes over bench.py --no-cpu --no-roofline --no-others --steps 40",
                  "note": "counters of a SEPARATE run of the same bench on another box of the pool: not the run this JSON line was printed by"}))
PY
cd $R && python3 tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json $O/pmc_traffic_per_kernel.txt > /dev/null && python3 tools/pmc_mfma.py $O/sq $O/pmc_mfma_busy.json $O/pmc_sq_per_kernel.txt > /dev/null && python3 tools/prof_summary.py $O/stats x y > $O/step_summary.txt 2>&1
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $O/stats/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/stats $O/fetch $O/write $O/sq
ls -la $O
