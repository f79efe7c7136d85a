set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=20 > $O/gputest16.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputest16.log
python bench.py > $O/r05_bench_line_final_a.json 2> $O/final_a.err; cp bench_detail.json $O/r05_bench_detail_final_a.json; tail -c 300 $O/r05_bench_line_final_a.json
