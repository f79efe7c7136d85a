#!/bin/bash
# host synchronisation of padding-free execution: does it leave the GPU idle?
set -e
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 400 python tools/pack_sync_probe.py 40 > $O/pack_sync_probe.txt 2>&1 || { tail -20 $O/pack_sync_probe.txt; exit 1; }
cat $O/pack_sync_probe.txt | grep -v Warning
