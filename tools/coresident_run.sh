#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/cores
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 ${PROBE:-tools/coresident_probe.py} "$@" > $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
C=$(find $O/prof -name "*kernel_trace.csv" | tail -1)
python tools/coresident_trace.py $C > $O/coresident.txt
rm -rf $O/prof
cat $O/coresident.txt
