#!/bin/bash
# Same-box comparison of several environment settings on the default bench step:  tools/ab_multi.sh OUT ROUNDS "ENV1" "ENV2" ... [-- bench args]
O=$PWD/gpurun_out/$1; R=$2; shift 2
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ "$1" = "--" ] && shift
: > $O
for r in $(seq $R); do for E in "${ENVS[@]}"; do
  echo -n "$E : " >> $O
  ( export $E; timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline "$@" 2>>$O.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('median_ms_per_step'))" >> $O ) || exit 1
done; done
cat $O
