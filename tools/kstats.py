#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as ms/step.  usage: kstats.py <dir> <steps> [top]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = list(csv.DictReader(open(f)))
for r in rows[:top]:
    print(f"{r['Name'][:70]:70s} {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms/step  n={int(r['Calls']) / steps:5.1f} avg {float(r['AverageNs']) / 1e3:7.1f} us")
print("total ms/step", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps)
