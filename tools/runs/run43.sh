#!/bin/bash
# host time of one C3 step (RoBERTa-base, mixed precision)
set -e
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 400 python tools/host_time_probe.py 40 bf16 roberta > $O/host_time_probe_c3.txt 2>&1 || { tail -20 $O/host_time_probe_c3.txt; exit 1; }
grep "^step" $O/host_time_probe_c3.txt
