#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_optim_gpu.py tests/test_unpad_gpu.py tests/test_ops_gpu.py -x -q -k "optim or adamw or presplit or plane or unpad" > $O/aw_tests.log 2>&1 || { tail -40 $O/aw_tests.log; exit 1; }
tail -2 $O/aw_tests.log
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $O/aw_line.json 2> $O/aw_line.err || { tail -20 $O/aw_line.err; exit 1; }
  python - $O/aw_line.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("fused AdamW planes:", d["value"], d["ms_per_step"], d.get("loss"))
PY
done
