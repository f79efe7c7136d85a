#!/usr/bin/env python
"""Turn the rocprofv3 PMC passes of tools/pmc_passes.sh into profiles/pmc_gemm.json + a text summary.

usage: tools/pmc_to_json.py gpurun_out/pmc_rXX profiles/ rXX [json name, default pmc_gemm.json]
Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE (KB) under-reports wide 16-B/lane streaming reads
by exactly 2x on gfx950 -> doubled; WRITE_SIZE (KB) is exact for 16-B/lane stores.  GRBM_GUI_ACTIVE is summed
over the 8 XCDs."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
json_name = sys.argv[4] if len(sys.argv) > 4 else "pmc_gemm.json"


def load(pass_dir):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(src, pass_dir, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("mtvaf::", "").replace("void ", "").replace("(GemmArgsX, P256SK)", "").replace("(GemmArgsX)", "").replace("(GemmArgsP)", "").replace("(GemmArgs)", "").replace("(ab::Args, int)", "").replace("(ab::Args)", "")
            c = agg[name][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    return {k: {c: v[0] / v[1] for c, v in cs.items()} | {"_n": max(v[1] for v in cs.values())} for k, cs in agg.items()}


p1, p2, p3, p4 = load("p1"), load("p2"), load("p3"), load("p4")
out, lines = {}, []
for name in sorted(p1, key=lambda k: -p1[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * p1[k]["_n"]):
    if "gemm" not in name and "attn" not in name:
        continue
    a = p1[name]
    fetch_kb = p2.get(name, {}).get("FETCH_SIZE")
    write_kb = p3.get(name, {}).get("WRITE_SIZE")
    gui = a.get("GRBM_GUI_ACTIVE", 0) / 8.0
    rec = {"dispatches": a["_n"],
           "mfma_busy_frac": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024), 4) if gui else None,
           "wait_any_frac": round(a.get("SQ_WAIT_ANY", 0) / a["SQ_WAVE_CYCLES"], 4) if a.get("SQ_WAVE_CYCLES") else None,
           "wait_inst_any_frac": round(a.get("SQ_WAIT_INST_ANY", 0) / a["SQ_WAVE_CYCLES"], 4) if a.get("SQ_WAVE_CYCLES") else None,
           "mfma_flop_per_launch": (a.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) + a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0)) * 512,
           "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
           "traffic_bytes_per_launch": None if fetch_kb is None or write_kb is None else (2 * fetch_kb + write_kb) * 1024,
           "lds_bank_conflict_cycles": p4.get(name, {}).get("SQ_LDS_BANK_CONFLICT")}
    out[name] = rec
    lines.append(f"{name}\n    " + "  ".join(f"{k}={v}" for k, v in rec.items()))
os.makedirs(dst, exist_ok=True)
out["_source"] = (f"profiles/{json_name}: rocprofv3 --pmc passes of round {tag} (tools/pmc_passes.sh, separate profiled runs of the same "
                  "bench command under MTVAF_DW_STREAM=0), reduced by tools/pmc_to_json.py")
json.dump(out, open(os.path.join(dst, json_name), "w"), indent=1)
open(os.path.join(dst, f"{tag}_pmc_summary{'' if json_name == 'pmc_gemm.json' else '_' + json_name.replace('pmc_gemm_', '').replace('.json', '')}.txt"), "w").write(
    "rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1` (tools/pmc_passes.sh); per-dispatch means.\n"
    "traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950 FETCH_SIZE correction for 16-B/lane streams).\n\n" + "\n".join(lines) + "\n")
print("\n".join(lines[:12]))
