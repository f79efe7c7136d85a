#!/bin/bash
# Same-box A/B of an environment switch on the default bench step:  tools/ab_env.sh OUT "ENV_A" "ENV_B" [bench args...]
# (alternates A B A B; prints sentences/s, ms/step per run)
O=$PWD/gpurun_out/$1; A=$2; B=$3; shift 3
: > $O
for v in A B A B; do
  if [ $v = A ]; then E=$A; else E=$B; fi
  echo "== $v: $E" >> $O
  ( export $E; timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline "$@" 2>>$O.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('median_ms_per_step'))" >> $O ) || exit 1
done
cat $O
