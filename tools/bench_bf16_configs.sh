run() { echo "== $*"; timeout -k 10 500 python bench.py --no-cpu-baseline --no-roofline --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('mfma_fraction_of_step'))"; }
run --model roberta --dtype bf16
run --batch 64 --dtype bf16
run --batch 128 --seq 512 --dtype bf16 --steps 5 --warmup 4
