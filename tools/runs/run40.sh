#!/bin/bash
# tile walk of the GROUPED pre-split launch (weight gradients): row-major against bands inside each product's grid
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for rep in 1 2; do
 for g in 0 4 2; do
  MTVAF_P16_WALK_GG=$g timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary > $O/l40.json 2> $O/l40.err || { tail -20 $O/l40.err; exit 1; }
  python - bench_detail.json "P16_WALK_GG=$g" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
g=[x for x in r["per_kernel"] if "true, true, true" in x["kernel"]]
print(sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], r["all_gemm_kernels"]["frac"], "grouped", g[0]["avg_us"] if g else None)
PY
 done
done
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "grouped_weight_gradients" 2>&1 | tail -2
