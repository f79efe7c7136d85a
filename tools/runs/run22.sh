#!/bin/bash
# pre-split operand path: parity (planes vs in-kernel split; oracle-facing suites in planes mode), whole step with / without
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_unpad_gpu.py -x -q > $O/planes_t1.log 2>&1 || { tail -40 $O/planes_t1.log; exit 1; }
tail -2 $O/planes_t1.log
for w in 1 0 1 0; do
  MTVAF_F32_PLANES=$w timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary > $O/pl_line_$w.json 2> $O/pl_line_$w.err || { tail -20 $O/pl_line_$w.err; exit 1; }
  python - $O/pl_line_$w.json $w <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("F32_PLANES", sys.argv[2], d["value"], d["ms_per_step"], r["frac"], r.get("kernel","")[:70], d.get("loss"))
PY
done
