r() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ.get('MTVAF_DW_STREAM_MIN_ROWS'), sys.argv[1:], d['ms_per_step'], d['median_ms_per_step'], d.get('fwd_bwd_without_optimizer',{}).get('ms_per_step'))" "$@"; }
python -m pytest tests/test_optim_gpu.py tests/test_parallel.py -x -q -m gpu 2>&1 | tail -2
r
r
r --dtype bf16 --model roberta
r --dtype bf16 --batch 64
r --batch 4 --seq 64 --aux 3 --steps 50
r --batch 4 --seq 64 --aux 3 --steps 50
export MTVAF_DW_STREAM_MIN_ROWS=0
r --batch 4 --seq 64 --aux 3 --steps 50
r --batch 4 --seq 64 --aux 3 --steps 50
