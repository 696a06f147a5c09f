#!/usr/bin/env python3
"""HBM-side bytes per launch of the matrix-core kernels from two rocprofv3 --pmc passes over bench.py
(FETCH_SIZE and WRITE_SIZE collected separately; FETCH_SIZE is in KB and, on gfx950, counts a wide coalesced
read at half its bytes -- doubled here, as MI355X_MICROARCH.md's HBM/rocprofv3 section prescribes).

  python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [summary.txt]
Keys match bench.py's kernel-instance names."""
import collections, csv, glob, json, os, re, sys


def load(d, counter):
    f = max(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("void rbvae::", "").split("(")[0]
        agg[(n, int(r.get("Grid_Size", 0) or 0))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(fetch):
    rd, wr = 2.0 * fetch[k] * 1024, write.get(k, 0.0) * 1024          # KB -> bytes, gfx950 read correction
    rows.append((k[0], k[1], nf[k], rd, wr))
per_inst = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, grid, n, rd, wr in rows:
    m = re.match(r"(gather_gemm_k|wgrad_gemm_k)<([^>]*)>", name)
    if name.startswith("wgrad_row_k"):
        p = per_inst["wgrad_row_k<3 taps per workgroup, K-split>"]
        p[0] += n; p[1] += rd * n; p[2] += wr * n
        continue
    if not m:
        continue
    a = [x.strip() for x in m.group(2).split(",")]
    if m.group(1) == "gather_gemm_k":
        key = "gather_gemm_k<%s, %s, %s, %s>" % (a[0], a[1], a[2], a[3])
        if len(a) > 5 and a[5] == "64":
            key = "gather_gemm_k<%s, %s, %s, %s, 64-row tile>" % (a[0], a[1], a[2], a[3])
        if a[3] == "1" or (a[1] == "2" and a[3] == "2"):
            key = "gather_gemm_k<%s, single/double buffer>" % a[0]
    else:
        key = "wgrad_gemm_k<%s, %s, K-split>" % (a[0], a[1])
    p = per_inst[key]
    p[0] += n; p[1] += rd * n; p[2] += wr * n
out = {k: {"read_bytes": round(v[1] / v[0]), "write_bytes": round(v[2] / v[0]), "bytes": round((v[1] + v[2]) / v[0]),
           "launches_sampled": v[0], "source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE, separate passes"}
       for k, v in per_inst.items()}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
if len(sys.argv) > 4:
    with open(sys.argv[4], "w") as f:
        f.write("kernel | grid threads | launches | HBM-side read MB (FETCH_SIZE x2) | write MB (WRITE_SIZE)\n")
        for name, grid, n, rd, wr in rows:
            f.write(f"{name[:60]:60s} {grid:8d} {n:5d} {rd / 1e6:10.2f} {wr / 1e6:10.2f}\n")
print(json.dumps(out, indent=1, sort_keys=True))
