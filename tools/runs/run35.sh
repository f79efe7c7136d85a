#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for rep in 1 2; do
 for nb in 512 768 1024 384; do
  MTVAF_LN_BWD_BLOCKS=$nb timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $O/l35.json 2> $O/l35.err || { tail -20 $O/l35.err; exit 1; }
  python - $O/l35.json "LN_BWD_BLOCKS=$nb" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"])
PY
 done
done
