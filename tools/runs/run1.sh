set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest1.log 2>&1; echo "pytest rc=$?"
tail -5 $O/gputest1.log
timeout -k 10 200 python tools/f32x3_bench.py 2432 --tiles > $O/x3_bench_2432.txt 2>&1; echo "x3bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/line1.json 2> $O/line1.err; echo "bench rc=$?"
tail -c 1500 $O/line1.json
