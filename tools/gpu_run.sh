#!/bin/bash
# scratch: one GPU call
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/gpu_suite.txt; rc=$?
cat gpurun_out/gpu_suite.txt
[ $rc = 0 ] && timeout -k 10 300 python __graft_entry__.py --smoke 2>&1 | tail -6 | cut -c1-400
