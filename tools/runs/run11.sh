set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
S="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
one() {
  python tools/attn_bench.py 2>&1 | sed "s/^/$1 /"
  for c in "" "--padded"; do
  python bench.py $S $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$c', d['value'], d['ms_per_step'], d['median_ms_per_step'])"
  done
}
python bench.py $S > /dev/null 2>&1   # (warm the box)
one group
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form=1 -DMTVAF_ATTN_XCD_GROUP=0 -c mtvaf_amd/csrc/attention.hip -o mtvaf_amd/lib/attention.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mtvaf_amd/lib/libmtvaf_hip.so mtvaf_amd/lib/*.o
one plain
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention or attn" > $O/gputest11.log 2>&1; echo "pytest (plain build) rc=$?"; tail -2 $O/gputest11.log
