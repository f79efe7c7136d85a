set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
for v in 0 1 2 3; do
  if [ $v = 0 ]; then export MTVAF_LIB=; else export MTVAF_LIB=$PWD/mtvaf_amd/lib_p256_$v/libmtvaf_hip.so; fi
  [ $v = 0 ] && unset MTVAF_LIB
  timeout -k 10 300 python tools/p256_bench.py 4864 38912 > gpurun_out/r6c/p256_v$v.txt 2>&1
  tail -2 gpurun_out/r6c/p256_v$v.txt
done
for v in 1 2 3; do
  export MTVAF_LIB=$PWD/mtvaf_amd/lib_p256_$v/libmtvaf_hip.so
  timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "bf16" > gpurun_out/r6c/tests_v$v.log 2>&1; tail -2 gpurun_out/r6c/tests_v$v.log
done
echo finished
