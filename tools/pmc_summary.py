#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv per kernel name: mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].replace("mtvaf::", "").replace("void ", "")[:70]
    c = agg[name][r["Counter_Name"]]
    c[0] += float(r["Counter_Value"])
    c[1] += 1
only = sys.argv[2] if len(sys.argv) > 2 else "gemm"
for name, cs in sorted(agg.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
    if only not in name:
        continue
    n = max(v[1] for v in cs.values())
    print(f"{name}  (dispatches {n})")
    vals = {k: v[0] / v[1] for k, v in cs.items()}
    for k, v in sorted(vals.items()):
        print(f"    {k:32s} {v:16.1f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
        # GRBM_GUI_ACTIVE is summed over 8 XCDs; 1024 SIMDs on the chip
        gui = vals["GRBM_GUI_ACTIVE"] / 8.0
        print(f"    -> MFMA busy fraction = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs) = "
              f"{vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):.3f}")
    if "SQ_WAVE_CYCLES" in vals:
        w = vals["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in vals:
                print(f"    -> {k}/WAVE_CYCLES = {vals[k] / w:.3f}")
