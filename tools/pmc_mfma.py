#!/usr/bin/env python3
"""Matrix-core busy fraction and wave occupancy per matrix-core kernel instance from one rocprofv3 --pmc pass over
bench.py (SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES
SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY + GRBM_GUI_ACTIVE).

  python tools/pmc_mfma.py <pmc_dir> <out.json> [summary.txt]

Derived figures (MI355X_MICROARCH.md, "Per-instruction cycle constants" and "DVFS give-back"):
  kernel_cycles      = GRBM_GUI_ACTIVE / 8                      (the counter is summed over the 8 XCDs)
  mfma_busy_frac     = SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles * 1024 SIMDs)
                       (the counter counts cycles, = 16 per v_mfma_f32_16x16x32_bf16, summed over every SIMD)
  mfma_issue_frac    = SQ_INSTS_MFMA * 16 / (kernel_cycles * 1024)     (bf16 16x16x32: 16 cycles per instruction and SIMD)
  waves_per_cu       = SQ_WAVE_CYCLES * 4 / (kernel_cycles * 256 CUs)   (quad-cycles -> cycles; time-averaged resident waves)
Keys match bench.py's kernel-instance names."""
import collections, csv, glob, json, os, re, sys

d = sys.argv[1]
f = max(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void rbvae::", "").split("(")[0]
    agg[(n, int(r.get("Grid_Size", 0) or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))


def inst_key(name):
    if name.startswith("wgrad_row_k"):
        return "wgrad_row_k<3 taps per workgroup, K-split>"
    m = re.match(r"(gather_gemm_k|wgrad_gemm_k)<([^>]*)>", name)
    if not m:
        return None
    a = [x.strip() for x in m.group(2).split(",")]
    if m.group(1) == "gather_gemm_k":
        if a[3] == "1" or (a[1] == "2" and a[3] == "2"):
            return "gather_gemm_k<%s, single/double buffer>" % a[0]
        if len(a) > 5 and a[5] == "64":
            return "gather_gemm_k<%s, %s, %s, %s, 64-row tile>" % (a[0], a[1], a[2], a[3])
        return "gather_gemm_k<%s, %s, %s, %s>" % (a[0], a[1], a[2], a[3])
    return "wgrad_gemm_k<%s, %s, K-split>" % (a[0], a[1])


per = collections.defaultdict(lambda: collections.defaultdict(float))
lines = []
for (name, grid), cs in sorted(agg.items()):
    n = len(next(iter(cs.values())))
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    lines.append(f"{name[:56]:56s} grid {grid:8d} x{n:4d} " + " ".join(f"{c}={v:.4g}" for c, v in sorted(avg.items())))
    k = inst_key(name)
    if k:
        for c, v in cs.items():
            per[k][c] += sum(v)
        per[k]["_launches"] += n
out = {}
for k, c in per.items():
    n = c["_launches"]
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    e = {"launches_sampled": int(n), "kernel_cycles_avg": round(cyc / n),
         "source": "rocprofv3 --pmc SQ_* + GRBM_GUI_ACTIVE pass over bench.py (tools/pmc_mfma.py)"}
    if cyc > 0:
        e["mfma_busy_frac"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4)
        e["mfma_issue_frac"] = round(c.get("SQ_INSTS_MFMA", 0.0) * 16 / (cyc * 1024), 4)
        e["waves_per_cu"] = round(c.get("SQ_WAVE_CYCLES", 0.0) * 4 / (cyc * 256), 2)
    for raw in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES",
                "SQ_WAVES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"):
        if raw in c:
            e[raw + "_per_launch"] = round(c[raw] / n)
    out[k] = e
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(lines) + "\n")
print(json.dumps(out, indent=1, sort_keys=True))
