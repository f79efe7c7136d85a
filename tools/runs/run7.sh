set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
S="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
for bg in 128 32 64 256 512; do
  MTVAF_ADAMW_BG_BLOCKS=$bg python bench.py $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bg', $bg, d['value'], d['ms_per_step'], d['median_ms_per_step'])"
done
MTVAF_ADAMW_BG_MIN_ROWS=100000 python bench.py $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full-width', d['value'], d['ms_per_step'], d['median_ms_per_step'])"
MTVAF_ADAMW_BG_BLOCKS=128 python bench.py $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bg 128 again', d['value'], d['ms_per_step'], d['median_ms_per_step'])"
python bench.py $S --no-overlap-optimizer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no overlap', d['value'], d['ms_per_step'], d['median_ms_per_step'])"
