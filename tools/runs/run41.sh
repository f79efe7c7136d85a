#!/bin/bash
# split-K plan of the 768-wide pre-split products: two slabs from 8 k-tiles (default) or only from 32 / 80 (Wo unsplit; Wo and QKV dX unsplit)
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for rep in 1 2; do
 for g in 8 32 80; do
  MTVAF_P16_SPLIT_MIN_KT=$g timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary > $O/l41.json 2> $O/l41.err || { tail -20 $O/l41.err; exit 1; }
  python - bench_detail.json "P16_SPLIT_MIN_KT=$g" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], r["all_gemm_kernels"]["frac"], [(x["N"],x["K"],x["splits"],x["avg_us"]) for x in r["per_shape"] if x["N"]==768 and x["M"]==2432])
PY
 done
done
