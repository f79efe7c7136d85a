#!/bin/bash
# scratch: one GPU call
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "presplit" 2>&1 | tail -15
