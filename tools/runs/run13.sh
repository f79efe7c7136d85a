set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_unpad_gpu.py -m gpu -q -k "bf16" > $O/gputest13.log 2>&1; echo "pytest rc=$?"; tail -15 $O/gputest13.log
