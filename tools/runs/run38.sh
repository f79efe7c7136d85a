#!/bin/bash
# tile walk of the pre-split kernel (placement only): row-major against bands of G tile rows walked column by column
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for rep in 1 2; do
 for g in 0 4 2 8; do
  MTVAF_P16_WALK_G=$g timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary > $O/l38.json 2> $O/l38.err || { tail -20 $O/l38.err; exit 1; }
  python - $O/l38.json "P16_WALK_G=$g" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print(sys.argv[2], d["value"], d["ms_per_step"], r["kernel"][-30:], r["avg_launch_us"], r["frac"], r["all_gemm_frac"])
PY
 done
done
