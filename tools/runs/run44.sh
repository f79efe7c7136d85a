#!/bin/bash
# host time of the launch-bound configurations: C1 (fp32, bs 4 / S 64) and C3 (RoBERTa-base, mixed precision)
set -e
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 400 python tools/host_time_probe.py 40 fp32 bert 4 64 > $O/host_time_probe_c1.txt 2>&1 || { tail -20 $O/host_time_probe_c1.txt; exit 1; }
grep "^step" $O/host_time_probe_c1.txt
timeout -k 10 400 python tools/host_time_probe.py 40 bf16 roberta > $O/host_time_probe_c3.txt 2>&1 || { tail -20 $O/host_time_probe_c3.txt; exit 1; }
grep "^step" $O/host_time_probe_c3.txt
