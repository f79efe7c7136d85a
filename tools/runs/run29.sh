#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for i in 1 2 3; do
 for w in 1 0; do
  MTVAF_DW_JOBS=$w timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $O/l29.json 2> $O/l29.err || { tail -20 $O/l29.err; exit 1; }
  python - $O/l29.json "DW_JOBS=$w" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"], d.get("loss"))
PY
 done
done
