set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "stream_k or dw_group" > gpurun_out/r6d/tests.log 2>&1; tail -2 gpurun_out/r6d/tests.log
for k in 0 1 0 1; do
  MTVAF_P256_SK_KMAJOR=$k timeout -k 10 300 python tools/p256_bench.py 2560 4864 38912 2>&1 | grep "four dW" | sed "s/^/kmajor=$k /" >> gpurun_out/r6d/dwgroup.txt
done
cat gpurun_out/r6d/dwgroup.txt
for k in 0 1; do
  MTVAF_P256_SK_KMAJOR=$k timeout -k 10 300 python bench.py --dtype bf16 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r6d/c4_k$k.json 2> gpurun_out/r6d/c4_k$k.err
  MTVAF_P256_SK_KMAJOR=$k timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --seq 512 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r6d/c5_k$k.json 2> gpurun_out/r6d/c5_k$k.err
done
grep -o '"value": [0-9.]*\|"frac": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/r6d/c4_k*.json gpurun_out/r6d/c5_k*.json
echo finished
