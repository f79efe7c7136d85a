#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -k "presplit" > $O/p16g_tests.log 2>&1 || { tail -40 $O/p16g_tests.log; exit 1; }
tail -2 $O/p16g_tests.log
timeout -k 10 600 python tools/f32p_bench.py 2432 4096 2>/dev/null > $O/f32p_bench_grouped.txt; grep "grouped\|four" $O/f32p_bench_grouped.txt
