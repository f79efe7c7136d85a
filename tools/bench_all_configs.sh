# Re-measures every BASELINE configuration.  The S=512 lines need 4 warm-up steps:
# the first steps of an 84-GiB working set spend seconds in hipMalloc until the caching allocator is warm.
run() { echo "== $*"; timeout 600 python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('mfma_fraction_of_step'))"; }
run --batch 4 --seq 64 --aux 3
run --full-length
run --model roberta --dtype bf16
run --batch 64 --dtype bf16
run --batch 64
run --batch 128 --seq 512 --steps 5 --warmup 4
run --batch 128 --seq 512 --dtype bf16 --steps 5 --warmup 4
run --dtype bf16
run --optimizer reference
run --batch 4 --seq 64 --aux 3 --graph
run --dtype bf16 --graph
run --batch 4 --seq 64 --aux 3 --dtype bf16
run --unpad
run --unpad --batch 64
run --unpad --dtype bf16 --model roberta
run --unpad --dtype bf16 --batch 64
run --unpad --batch 128 --seq 512 --steps 5 --warmup 4
run --unpad --batch 128 --seq 512 --dtype bf16 --steps 5 --warmup 4
