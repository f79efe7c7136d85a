#!/bin/bash
# where the pre-split path starts to pay: step time at smaller batches with / without it
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for b in 12 16 24; do
 for w in 1 0 1 0; do
  MTVAF_F32_PLANES=$w MTVAF_F32_PLANES_MIN_ROWS=256 timeout -k 10 300 python bench.py --batch $b --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $O/l32.json 2> $O/l32.err || { tail -20 $O/l32.err; exit 1; }
  python - $O/l32.json "batch $b F32_PLANES=$w" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"], d.get("real_token_rows"))
PY
 done
done
