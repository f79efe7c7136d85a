#!/bin/bash
# host time of one default step, by segment and by function
set -e
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 400 python tools/host_time_probe.py 40 > $O/host_time_probe.txt 2>&1 || { tail -20 $O/host_time_probe.txt; exit 1; }
grep "^step" $O/host_time_probe.txt
