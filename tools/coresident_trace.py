"""Reads the rocprofv3 --kernel-trace CSV of tools/coresident_probe.py: for every small kernel, its duration alone and while a
grouped weight-gradient launch (gemm_f32p16w_kernel<true, true, true...>) was running for the kernel's whole duration."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].replace("void ", "").replace("mtvaf::", "").split("(")[0][:70]
gem = [r for r in rows if r["n"].startswith("gemm_f32p16w_kernel<true, true, true")]
res = collections.defaultdict(lambda: [[], [], []])
for r in rows:
    if r["n"].startswith("gemm_f32p16w") or "split_planes" in r["n"]:
        continue
    full = any(g["s"] <= r["s"] and g["e"] >= r["e"] for g in gem)
    part = any(min(g["e"], r["e"]) > max(g["s"], r["s"]) for g in gem)
    res[r["n"]][0 if full else (2 if part else 1)].append((r["e"] - r["s"]) / 1e3)
print(f"{'kernel':70s} | alone: n, median us | inside a grouped-dW launch: n, median us | partly beside: n")
med = lambda v: sorted(v)[len(v) // 2] if v else float('nan')
for n, (inside, alone, part) in sorted(res.items(), key=lambda kv: -med(kv[1][1] or [0])):
    if len(alone) + len(inside) < 4:
        continue
    print(f"{n:70s} | {len(alone):4d} {med(alone):8.1f} | {len(inside):4d} {med(inside):8.1f} | {len(part):4d}")
print(f"grouped dW launches: {len(gem)}, median {med([(g['e'] - g['s']) / 1e3 for g in gem]):.1f} us")
