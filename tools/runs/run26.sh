#!/bin/bash
# the GPU suite in the non-default modes: $1 = environment assignment(s)
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
for mode in "$@"; do
  tag=$(echo "$mode" | tr '= ' '__')
  ( export $mode; timeout -k 10 560 python -m pytest tests -x -q -m gpu > $O/gpu_suite_$tag.log 2>&1 ) || { tail -30 $O/gpu_suite_$tag.log; exit 1; }
  echo "$mode: $(tail -1 $O/gpu_suite_$tag.log)"
done
