#!/bin/bash
# Are the `__amd_rocclr_copyBuffer` dispatches of a profiled bench run per-step work or one-time set-up (parameters moved to
# the device, the per-layer flat buffers filled once)?  Profiles the same command with 3 and with 13 timed steps: per-step
# copies would grow by 10 x (copies per step), set-up copies stay the same.   tools/count_copies.sh [bench args...]
export TMPDIR=/tmp
O=$PWD/gpurun_out/copies
mkdir -p $O
for K in 3 13; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$K -- python3 bench.py --steps $K --warmup 2 --no-cpu-baseline --no-roofline --no-secondary "$@" > $O/k$K.log 2>&1
  f=$(find $O/k$K -name "*kernel_stats.csv" | tail -1)
  echo "steps=$K (+2 warm-up): $(grep -h copyBuffer $f | awk -F, '{print "copyBuffer calls", $2, "total ns", $3}')"
  rm -rf $O/k$K
done
