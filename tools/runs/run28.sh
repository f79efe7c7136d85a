#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_unpad_gpu.py tests/test_properties_gpu.py -x -q -k "presplit or plane or unpad or dropout_res_ln or layernorm or properties or padding" > $O/t28.log 2>&1 || { tail -40 $O/t28.log; exit 1; }
tail -2 $O/t28.log
for i in 1 2; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $O/l28.json 2> $O/l28.err || { tail -20 $O/l28.err; exit 1; }
  python - $O/l28.json "default" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"], d.get("loss"))
PY
  MTVAF_LN_FINISH_RG32=0 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $O/l28.json 2> $O/l28.err || { tail -20 $O/l28.err; exit 1; }
  python - $O/l28.json "LN_FINISH_RG32=0" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"], d.get("loss"))
PY
  MTVAF_ADAMW_BG_MIN_ROWS=100000000 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $O/l28.json 2> $O/l28.err || { tail -20 $O/l28.err; exit 1; }
  python - $O/l28.json "AdamW full width" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"], d.get("loss"))
PY
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --full-length --no-cpu-baseline --no-secondary > $O/l28_full.json 2> $O/l28.err || { tail -20 $O/l28.err; exit 1; }
python - $O/l28_full.json "full-length" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"][:60])
PY
