#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: top kernels, ms per step."""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {steps:g} steps -> {tot/1e6/steps:.2f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:60]
    print(f"{n:60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  per-step {float(r['TotalDurationNs'])/1e6/steps:7.2f} ms  {float(r['Percentage']):5.1f}%")
