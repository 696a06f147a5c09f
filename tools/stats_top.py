"""Top kernels of a rocprofv3 --stats run: python tools/stats_top.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f'{r["Name"][:84]:84s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms {100 * float(r["TotalDurationNs"]) / tot:5.1f}% avg {float(r["AverageNs"]) / 1e3:8.1f} us')
