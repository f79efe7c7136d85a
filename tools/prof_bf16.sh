#!/bin/bash
# Serialised rocprofv3 kernel stats of the bf16-compute mode at the C3 / C4 shapes (+ the fp32 headline for reference).
# usage: tools/prof_bf16.sh <tag>   -> gpurun_out/<tag>_{c3,c4,c2}/...
export TMPDIR=/tmp
export MTVAF_DW_STREAM=0
R=$PWD
T=${1:-r02}
A="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-optimizer"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_c3 -- python3 bench.py $A --model roberta --dtype bf16 > $R/gpurun_out/${T}_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_c4 -- python3 bench.py $A --batch 64 --dtype bf16 > $R/gpurun_out/${T}_c4.log 2>&1
find $R/gpurun_out/${T}_c3 $R/gpurun_out/${T}_c4 -name "*kernel_trace.csv" -delete
