run() { echo "== $*"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['loss'])"; }
MTVAF_F32_SPLIT=1 run --unpad
MTVAF_F32_SPLIT=0 run --unpad
MTVAF_F32_SPLIT=1 run --batch 64
MTVAF_F32_SPLIT=0 run --batch 64
