#!/bin/bash
# One GPU call: kernel trace of the default bench step, every kernel of one step with its start time, duration and queue.
#   tools/step_trace.sh <tag> [bench args...]
export TMPDIR=/tmp
T=${1:-trace}; shift
R=$PWD
O=$R/gpurun_out/$T
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary "$@" > $O/bench.log 2>&1 || exit 1
C=$(find $O/prof -name "*kernel_trace.csv" | tail -1)
python tools/overlap_probe.py $C 1 0 > $O/step_kernels.txt
python tools/timeline.py $C 1 > $O/timeline.txt 2>&1
rm -rf $O/prof
tail -3 $O/timeline.txt
