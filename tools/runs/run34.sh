#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for rep in 1 2; do
 for nb in 128 64 192 256 512; do
  MTVAF_ADAMW_BG_BLOCKS=$nb timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $O/l34.json 2> $O/l34.err || { tail -20 $O/l34.err; exit 1; }
  python - $O/l34.json "ADAMW_BG_BLOCKS=$nb" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"])
PY
 done
done
