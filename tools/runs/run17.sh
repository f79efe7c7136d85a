set -o pipefail
export TMPDIR=/tmp
S="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
python bench.py $S > /dev/null 2>&1
for c in 512 768 1024 2048; do
  MTVAF_LN_BWD_BLOCKS=$c python tools/ln_time.py 2>/dev/null | sed "s/^/cap $c: /"
  MTVAF_LN_BWD_BLOCKS=$c python bench.py $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap $c', d['value'], d['ms_per_step'], d['median_ms_per_step'])"
done
MTVAF_LN_BWD_BLOCKS=512 python bench.py $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap 512 again', d['value'], d['ms_per_step'], d['median_ms_per_step'])"
