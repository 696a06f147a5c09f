#!/bin/bash
# LDS counters of the GEMM kernels in the default bench (bank conflicts vs active LDS cycles): gpurun_out/pmc_lds.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_lds; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o l -- python3 $R/bench.py --no-cpu --no-roofline --steps 30 > $O/run.log 2>&1 || exit 1
cd $R && python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void rbvae::", "").split("(")[0]
    agg[(n, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for (n, g), cs in sorted(agg.items()):
    if "gemm" not in n and "lstm" not in n and "conv" not in n:
        continue
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    act = m.get("SQ_LDS_IDX_ACTIVE", 0.0)
    out.append(f"{n[:52]:52s} grid {g:>8s}  LDS active {act:12.0f}  bank conflict {m.get('SQ_LDS_BANK_CONFLICT', 0):12.0f} "
               f"({100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / max(act, 1):5.1f} %)  addr conflict {m.get('SQ_LDS_ADDR_CONFLICT', 0):10.0f}  "
               f"insts {m.get('SQ_INSTS_LDS', 0):10.0f}  data fifo full {m.get('SQ_LDS_DATA_FIFO_FULL', 0):10.0f}  cmd fifo full {m.get('SQ_LDS_CMD_FIFO_FULL', 0):10.0f}  GRBM {m.get('GRBM_GUI_ACTIVE', 0):9.0f}")
open("gpurun_out/pmc_lds.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $O
