cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6l
timeout -k 10 900 python -m pytest tests/test_unpad_gpu.py tests/test_configs_gpu.py tests/test_model_gpu.py -m gpu -x -q > gpurun_out/r6l/tests.log 2>&1; tail -3 gpurun_out/r6l/tests.log
for rep in 1 2; do for k in 0 1; do
  MTVAF_ATTN_ORDER=$k timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r6l/c2_o${k}_$rep.json 2> gpurun_out/r6l/c2_o${k}_$rep.err
done; done
for k in 0 1; do
  MTVAF_ATTN_ORDER=$k timeout -k 10 300 python bench.py --steps 20 --warmup 5 --dtype bf16 --batch 64 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r6l/c4_o$k.json 2> gpurun_out/r6l/c4_o$k.err
  MTVAF_ATTN_ORDER=$k timeout -k 10 300 python bench.py --steps 20 --warmup 5 --dtype bf16 --model roberta --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r6l/c3_o$k.json 2> gpurun_out/r6l/c3_o$k.err
done
grep -o '"value": [0-9.]*' gpurun_out/r6l/*.json
