#!/bin/bash
# whole GPU suite in the default mode (pre-split operand path on), smoke, default bench line
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_suite_planes.log 2>&1 || { tail -40 $O/gpu_suite_planes.log; exit 1; }
tail -2 $O/gpu_suite_planes.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > $O/line_planes.json 2> $O/line_planes.err || { tail -20 $O/line_planes.err; exit 1; }
python - $O/line_planes.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["kernel"], r.get("avg_launch_us"), r.get("launches_per_step"), r.get("traffic"))
print(d["config"]["workload"][-160:])
print(len(open(sys.argv[1]).read()), "bytes")
PY
