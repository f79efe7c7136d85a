#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -k "presplit or plane_images or attention" > $O/p16e_tests.log 2>&1 || { tail -40 $O/p16e_tests.log; exit 1; }
tail -2 $O/p16e_tests.log
timeout -k 10 600 python -m pytest tests/test_unpad_gpu.py -x -q > $O/planes_t1.log 2>&1 || { tail -40 $O/planes_t1.log; exit 1; }
tail -2 $O/planes_t1.log
for w in "1 1" "1 0" "0 1" "1 1" "1 0" "0 1"; do
  set -- $w
  MTVAF_F32_PLANES=$1 MTVAF_ATTN_PLANES=$2 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $O/pl_line.json 2> $O/pl_line.err || { tail -20 $O/pl_line.err; exit 1; }
  python - $O/pl_line.json "$w" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("F32_PLANES ATTN_PLANES =", sys.argv[2], d["value"], d["ms_per_step"], d.get("loss"))
PY
done
