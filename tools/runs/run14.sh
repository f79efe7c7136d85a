set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parallel.py -m gpu -q > $O/gputest14.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gputest14.log
