set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 200 python tools/f32x3_bench.py 2432 --tiles > $O/x3_bench_2432.txt 2>&1; echo "x3bench rc=$?"
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=40 > $O/gputest2.log 2>&1; echo "pytest rc=$?"
tail -45 $O/gputest2.log
