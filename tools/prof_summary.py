#!/usr/bin/env python
"""Per-step summary of a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv)."""
import csv
import sys

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'ms/step':>9} {'calls/step':>10} {'avg us':>9} {'%':>6}  kernel")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = r["Name"].replace("mtvaf::", "").replace("void ", "")[:100]
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:9.3f} {int(r['Calls']) / steps:10.1f} "
          f"{float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):6.1f}  {n}")
print(f"total kernel time per step: {tot / 1e6 / steps:.3f} ms")
