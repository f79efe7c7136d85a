set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "presplit or tile64" > $O/gputest5.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gputest5.log
timeout -k 10 400 python tools/f32p_bench.py 4096 2432 > $O/r05_f32p_bench.txt 2>&1; echo "rc=$?"
grep -v natural $O/r05_f32p_bench.txt | cut -c1-300
