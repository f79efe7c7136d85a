#!/usr/bin/env python
"""Per-step view of a rocprofv3 --kernel-trace CSV: the kernels between two consecutive CRF forward launches (one training
step), their busy time and the time per kernel name.  usage: trace_step.py <kernel_trace.csv> [step index]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "crf_fwd_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 15
seg = rows[idx[k]:idx[k + 1]]
span = int(rows[idx[k + 1]]["Start_Timestamp"]) - int(rows[idx[k]]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"step {k}: {len(seg)} kernels, span {span / 1e6:.3f} ms (profiled), sum of kernel durations {busy / 1e6:.3f} ms")
d = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r["Kernel_Name"].replace("void ", "").replace("mtvaf::", "")[:78]
    d[n][0] += 1
    d[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{v[1] / 1e3:8.1f} us {v[0]:4d} x {v[1] / v[0] / 1e3:6.1f}  {n}")
