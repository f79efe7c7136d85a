#!/bin/bash
# bf16 planner: 128-wide tiles where they save a round (4864 packed rows) -- tests + C3 / C4 lines
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py -x -q -k "bf16" > $O/bf16_tests.log 2>&1 || { tail -30 $O/bf16_tests.log; exit 1; }
tail -2 $O/bf16_tests.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --dtype bf16 --batch 64 --no-cpu-baseline --no-secondary > $O/c4_new_$i.json 2> $O/c4_new.err || { tail $O/c4_new.err; exit 1; }
python bench.py --steps 20 --warmup 5 --dtype bf16 --model roberta --no-cpu-baseline --no-secondary > $O/c3_new_$i.json 2> $O/c3_new.err || { tail $O/c3_new.err; exit 1; }
python - $O/c4_new_$i.json $O/c3_new_$i.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f)); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("kernel","")[:60])
PY
done
