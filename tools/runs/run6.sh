set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=20 > $O/gputest6.log 2>&1; echo "pytest rc=$?"
tail -12 $O/gputest6.log
