set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
TAG=${1:-base}
timeout -k 10 300 python tools/bf16x_bench.py > $O/bf16x_bench_$TAG.txt 2>&1; echo "rc=$?"
tail -30 $O/bf16x_bench_$TAG.txt | cut -c1-250
for c in "--dtype bf16 --model roberta" "--dtype bf16 --batch 64"; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])"
done
