#!/bin/bash
# Same-box comparison of bench argument sets (each quoted):  tools/ab_args.sh OUT ROUNDS "ARGS1" "ARGS2" ...   (ENV=val prefixes allowed)
O=$PWD/gpurun_out/$1; R=$2; shift 2
: > $O
for r in $(seq $R); do for A in "$@"; do
  echo -n "$A : " >> $O
  E=""; ARGS=""
  for w in $A; do case $w in *=*) E="$E $w";; *) ARGS="$ARGS $w";; esac; done
  ( [ -n "$E" ] && export $E; timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline $ARGS 2>>$O.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('median_ms_per_step'))" >> $O ) || exit 1
done; done
cat $O
