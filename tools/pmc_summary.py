#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv)."""
import csv, glob, sys, collections
d = sys.argv[1]
import os
f = max(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"].replace("void rbvae::", "").split("(")[0][:40]
    key = (n, r.get("Grid_Size", ""))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in key[0]:
        continue
    print(key[0], "grid", key[1], " ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(cs.items())))
