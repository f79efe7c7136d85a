"""One training step on the GPU's clock, from a rocprofv3 --kernel-trace CSV (reduced to Queue_Id, Kernel_Name, Start / End;
gzip accepted): span, union of busy time, idle gaps, per-queue kernel time, the largest gaps and what runs around them.

    python tools/timeline.py trace.csv[.gz] [step-index-from-the-end, default 1]
"""
import collections
import csv
import gzip
import sys

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
op = gzip.open if path.endswith(".gz") else open
rows = list(csv.DictReader(op(path, "rt")))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].replace("void ", "").replace("mtvaf::", "")
rows.sort(key=lambda r: r["s"])
emb = [i for i, r in enumerate(rows) if r["n"].startswith("ln_fwd_kernel<1>")]  # the embedding LayerNorm opens a step
a, b = emb[-1 - back], emb[-back]
step = rows[a:b]
t0 = step[0]["s"]
print(f"step: {len(step)} kernels, {(rows[b]['s'] - t0) / 1e6:.3f} ms from its first kernel to the next step's first")
ev = sorted((r["s"], r["e"], r["n"]) for r in step)
busy, gaps = 0, []
cs, ce, last = ev[0][0], ev[0][1], ev[0][2]
for s, e, n in ev[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, ce - t0, last, n))
        cs, ce, last = s, e, n
    elif e > ce:
        ce, last = e, n
busy += ce - cs
print(f"union of kernel time {busy / 1e6:.3f} ms, sum of kernel time {sum(r['e'] - r['s'] for r in step) / 1e6:.3f} ms, "
      f"idle inside the step {sum(g[0] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps")
q = collections.Counter()
qn = collections.Counter()
for r in step:
    q[r["Queue_Id"]] += r["e"] - r["s"]
    qn[r["Queue_Id"]] += 1
print("per queue:", {k: (qn[k], round(v / 1e6, 3)) for k, v in q.items()})
hist = collections.Counter()
for g in gaps:
    hist[min(int(g[0] / 1000), 20)] += g[0]
print("idle by gap length (us: ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
print("largest gaps (us, at ms, after -> before):")
for g in sorted(gaps, reverse=True)[:15]:
    print(f"  {g[0] / 1e3:7.1f} us at {g[1] / 1e6:6.2f} ms   {g[2][:50]} -> {g[3][:50]}")
# where the main-queue kernels wait on each other: gap between consecutive kernels of the busiest queue
mainq = q.most_common(1)[0][0]
mk = [r for r in step if r["Queue_Id"] == mainq]
gsum = collections.Counter()
gcnt = collections.Counter()
for x, y in zip(mk, mk[1:]):
    d = y["s"] - x["e"]
    if d > 0:
        gsum[x["n"][:44]] += d
        gcnt[x["n"][:44]] += 1
print(f"gaps between consecutive kernels of queue {mainq} ({len(mk)} kernels), by the kernel before the gap:")
for k, v in gsum.most_common(12):
    print(f"  {v / 1e6:6.3f} ms in {gcnt[k]:3d} gaps  after {k}")
print(f"  total {sum(gsum.values()) / 1e6:.3f} ms")
