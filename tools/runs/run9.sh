set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
one() {
  TAG=$1
  timeout -k 10 300 python tools/bf16x_bench.py 4096 2432 > $O/bf16x_bench_$TAG.txt 2>&1; echo "rc=$?"
  grep "auto\|12 products" $O/bf16x_bench_$TAG.txt | cut -c1-200
  for c in "--dtype bf16 --model roberta" "--dtype bf16 --batch 64" "--dtype bf16 --model roberta --padded"; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', '$c', d['value'], d['ms_per_step'])"
  done
}
one spread
# the same on the same box with the reads as bursts (round 4's schedule): rebuild only gemm_bf16x.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DMTVAF_BF16X_SPREAD=0 -c mtvaf_amd/csrc/gemm_bf16x.hip -o mtvaf_amd/lib/gemm_bf16x.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mtvaf_amd/lib/libmtvaf_hip.so mtvaf_amd/lib/*.o
one burst
