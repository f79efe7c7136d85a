#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
run() {  # label, env, args...
  local label=$1 envv=$2; shift 2
  ( export $envv; timeout -k 10 300 python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/l33.json 2> $O/l33.err ) || { tail -20 $O/l33.err; exit 1; }
  python - $O/l33.json "$label" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"])
PY
}
for rep in 1 2; do
  run "batch 8  planes" "MTVAF_F32_PLANES=1 MTVAF_F32_PLANES_MIN_ROWS=128" --batch 8
  run "batch 8  off   " "MTVAF_F32_PLANES=0" --batch 8
  run "batch 4  planes" "MTVAF_F32_PLANES=1 MTVAF_F32_PLANES_MIN_ROWS=128" --batch 4
  run "batch 4  off   " "MTVAF_F32_PLANES=0" --batch 4
  run "C1 (4x64) planes" "MTVAF_F32_PLANES=1 MTVAF_F32_PLANES_MIN_ROWS=128" --batch 4 --seq 64 --aux 3
  run "C1 (4x64) off   " "MTVAF_F32_PLANES=0" --batch 4 --seq 64 --aux 3
done
