#!/bin/bash
# scratch: one GPU call
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$PWD/gpurun_out/padded_ab_prev.txt
: > $O
for v in prev cur prev cur; do
  D=$GRAFT_REPO_ROOT
  [ $v = prev ] && D=$GRAFT_REPO_ROOT/_prev
  echo "== $v" >> $O
  ( cd $D && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --padded --no-cpu-baseline --no-secondary 2>$GRAFT_REPO_ROOT/gpurun_out/wide_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'][:44], r['avg_launch_us'], r['frac'], r.get('all_gemm_frac'), d.get('mfma_fraction_of_step_executed'))" >> $O ) || exit 1
done
cat $O
