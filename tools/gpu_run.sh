#!/bin/bash
# scratch: one GPU call
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/ovp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovp -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/ovp.log 2>&1 || exit 1
T=$(find gpurun_out/ovp -name "*kernel_trace.csv" | tail -1)
python tools/overlap_probe.py $T 2 8.5 > gpurun_out/overlap_tail.txt
python tools/overlap_probe.py $T 2 0 3.2 > gpurun_out/overlap_head.txt
rm -rf gpurun_out/ovp
echo done
