#!/bin/bash
# mixed-precision mode: background width of the per-layer AdamW launches (C3 = RoBERTa bs 32, C4 = bs 64)
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for rep in 1 2; do
 for nb in 128 256 192; do
  for cfg in "--model roberta" "--batch 64"; do
   MTVAF_ADAMW_BG_BLOCKS=$nb timeout -k 10 300 python bench.py --steps 30 --warmup 8 --dtype bf16 $cfg --no-cpu-baseline --no-secondary --no-roofline > $O/l42.json 2> $O/l42.err || { tail -20 $O/l42.err; exit 1; }
   python - $O/l42.json "ADAMW_BG_BLOCKS=$nb $cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"])
PY
  done
 done
done
