"""How the side-stream kernels (a layer's grouped weight gradients, its AdamW update) overlap with the main stream inside ONE training
step, from a rocprofv3 --kernel-trace CSV: per kernel name the mean duration in this (overlapped) run, and for every side-queue kernel
of the step when it ran, how long, and which main-queue kernels ran beside it.

    python tools/overlap_probe.py trace.csv [step-index-from-the-end, default 1]
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].replace("void ", "").replace("mtvaf::", "").split("(")[0]
rows.sort(key=lambda r: r["s"])
emb = [i for i, r in enumerate(rows) if r["n"].startswith("ln_fwd_kernel<1")]
a, b = emb[-1 - back], emb[-back]
step = rows[a:b]
t0 = step[0]["s"]
print(f"step: {len(step)} kernels, {(rows[b]['s'] - t0) / 1e6:.3f} ms")
q = collections.Counter()
for r in step:
    q[r["Queue_Id"]] += r["e"] - r["s"]
mainq = q.most_common(1)[0][0]
print("kernel time per queue (ms):", {k: round(v / 1e6, 3) for k, v in q.items()}, "main =", mainq)
dur = collections.defaultdict(list)
for r in step:
    dur[(r["Queue_Id"] == mainq, r["n"][:60])].append(r["e"] - r["s"])
print("mean duration per kernel in this run (us), main queue first:")
for (m, n), v in sorted(dur.items(), key=lambda kv: (-kv[0][0], -sum(kv[1]))):
    if sum(v) > 30e3:
        print(f"  {'main' if m else 'side'} {n:60s} x{len(v):3d} mean {sum(v) / len(v) / 1e3:7.1f}  total {sum(v) / 1e6:6.3f} ms")
side = [r for r in step if r["Queue_Id"] != mainq]
main = [r for r in step if r["Queue_Id"] == mainq]
print("side-queue kernels of the step: start (ms), duration (us), main-queue kernels beside them (overlap us)")
for r in side:
    if r["e"] - r["s"] < 20e3:
        continue
    ov = collections.Counter()
    for m in main:
        o = min(r["e"], m["e"]) - max(r["s"], m["s"])
        if o > 0:
            ov[m["n"][:28]] += o
    alone = (r["e"] - r["s"]) - sum(ov.values())
    print(f"  {(r['s'] - t0) / 1e6:6.3f} {(r['e'] - r['s']) / 1e3:7.1f} {r['n'][:34]:34s} | alone {max(alone, 0) / 1e3:5.1f} | " +
          ", ".join(f"{k} {v / 1e3:.0f}" for k, v in ov.most_common(4)))
last_main = max(m["e"] for m in main)
last_side = max(s_["e"] for s_ in side) if side else last_main
print(f"main queue ends at {(last_main - t0) / 1e6:.3f} ms, side queue at {(last_side - t0) / 1e6:.3f} ms; next step starts at {(rows[b]['s'] - t0) / 1e6:.3f} ms")
if len(sys.argv) > 3:  # dump every kernel of the step from this time (ms) on
    lo = float(sys.argv[3]) * 1e6
    hi = float(sys.argv[4]) * 1e6 if len(sys.argv) > 4 else 1e18
    print(f"all kernels of the step from {lo / 1e6} ms: start (ms), duration (us), queue, name")
    for r in step:
        if lo <= r["s"] - t0 <= hi:
            print(f"  {(r['s'] - t0) / 1e6:7.3f} {(r['e'] - r['s']) / 1e3:7.1f} q{r['Queue_Id']} {r['n'][:90]}  grid {r.get('Grid_Size', '')} wg {r.get('Workgroup_Size', '')}")
