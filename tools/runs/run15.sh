set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "slabs or dropout_res_ln" > $O/gputest15.log 2>&1; echo "pytest rc=$?"; tail -12 $O/gputest15.log
S="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
python bench.py $S > /dev/null 2>&1
for v in 1 0 1 0; do
MTVAF_LN_SLABS=$v python bench.py $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LN_SLABS=$v', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['loss'])"
done
