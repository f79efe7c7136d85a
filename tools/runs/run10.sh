set -o pipefail
export TMPDIR=/tmp
T=r05
O=$PWD/gpurun_out/$T
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_unpad_gpu.py tests/test_configs_gpu.py tests/test_ops_gpu.py -m gpu -q -k "bf16 or unpad" > $O/gputest10.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputest10.log
python bench.py --steps 20 --warmup 5 --dtype bf16 --model roberta --no-cpu-baseline --no-secondary > $O/${T}_bench_line_bf16_c3.json 2> $O/bench_c3.err
python bench.py --steps 20 --warmup 5 --dtype bf16 --batch 64 --no-cpu-baseline --no-secondary > $O/${T}_bench_line_bf16_c4.json 2> $O/bench_c4.err
python bench.py --steps 5 --warmup 4 --dtype bf16 --batch 128 --seq 512 --no-cpu-baseline --no-secondary > $O/${T}_bench_line_bf16_c5.json 2> $O/bench_c5.err
for f in c3 c4 c5; do python -c "import json; d=json.loads(open('$O/${T}_bench_line_bf16_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r['kernel'])"; done
