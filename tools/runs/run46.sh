#!/bin/bash
# priority of the weight-gradient stream on the pre-split path (0 = default, -1 = high)
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for rep in 1 2; do
 for pr in 0 -1; do
  MTVAF_DW_PRIORITY=$pr timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $O/l46.json 2> $O/l46.err || { tail -20 $O/l46.err; exit 1; }
  python - $O/l46.json "DW_PRIORITY=$pr" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"])
PY
 done
done
