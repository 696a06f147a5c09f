#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel time per step and the GEMM launches of one step."""
import csv
import glob
import sys
import collections

d = sys.argv[1]
import os
f = max(glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the bench's instrumented leg (eager, one stream, behind device-side sleeps) follows the timed graph-replayed
# steps: summarise the timed steps only
for i, r in enumerate(rows):
    if "spin_kernel" in r["Kernel_Name"] or r["Kernel_Name"].startswith("at::cuda::"):
        rows = rows[:i]
        break
names = [r["Kernel_Name"] for r in rows]
# one marker per step: the first kernel of a step (the batch gather), else the first conv
import os as _os
marks = [i for i, n in enumerate(names) if "gather_frames_k" in n]
if _os.environ.get("PROF_MARK"):                 # any other kernel that opens a step (e.g. im2col_fast_k for the LDM encoder)
    marks = [i for i, n in enumerate(names) if _os.environ["PROF_MARK"] in n]
if len(marks) < 12:
    marks = [i for i, n in enumerate(names) if "conv_first_fused_k<4, 0>" in n or "conv_first_fused_k<3, 0>" in n]
if len(marks) < 12:
    marks = [i for i, n in enumerate(names) if "combine_losses_k" in n]
if len(marks) < 3:
    sys.exit(f"prof_summary: {len(marks)} step marker kernel(s) found in {f} (PROF_MARK={_os.environ.get('PROF_MARK')!r}): "
             "need at least 3 steps to summarise")
# the last few marks before the cut are the eager warm-ups of the instrumented leg: skip up to 6, summarise up to 10 steps
skip = min(6, max(0, len(marks) - 3))
b_i = len(marks) - 1 - skip if skip else len(marks) - 1
a_i = max(0, b_i - 10)
a, b = marks[a_i], marks[b_i]
steps = b_i - a_i
agg = collections.OrderedDict()
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("void rbvae::", "").replace("rbvae::", "").split("(")[0][:48]
    t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c = agg.setdefault(n, [0, 0.0, 1e9, 0.0])
    c[0] += 1
    c[1] += t
    c[2] = min(c[2], t)
    c[3] = max(c[3], t)
span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 / steps
busy = sum(v[1] for v in agg.values()) / steps
print(f"step span {span:.1f} us, kernel busy {busy:.1f} us, kernels/step {(b - a) / steps:.0f}")
try:
    for n, (c, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:48s} x{c / steps:5.1f}  {t / steps:8.1f} us/step  avg {t / c:7.1f}  min {mn:7.1f} max {mx:7.1f}")
except BrokenPipeError:
    pass
one = (marks[b_i - 1], marks[b_i])           # the last summarised step
if len(sys.argv) > 2:
    a, b = one
    for r in rows[a:b]:
        n = r["Kernel_Name"]
        if "gemm" in n:
            t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            print(f"    {n.replace('void rbvae::', '')[:44]:44s} grid=({int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) {t:7.1f} us")
if len(sys.argv) > 3:
    # timeline of one step: start offset, duration, queue/stream of every kernel (shows side-stream overlap)
    a, b = one
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        n = r["Kernel_Name"].replace("void rbvae::", "").replace("rbvae::", "").split("(")[0][:44]
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        print(f"    t={s / 1e3:8.1f} +{(e - s) / 1e3:6.1f} us  q={r.get('Queue_Id', '?'):>3s} wgs={wg:5d}  {n}")
