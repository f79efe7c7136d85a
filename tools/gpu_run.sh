set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -s -k "config5_full_size_training or bf16" > gpurun_out/r6b/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r6b/tests.log
tail -5 gpurun_out/r6b/tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --model roberta --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r6b/c3_eager_$i.json 2> gpurun_out/r6b/c3_eager_$i.err
timeout -k 10 300 python bench.py --model roberta --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline --padded > gpurun_out/r6b/c3_padded_$i.json 2> gpurun_out/r6b/c3_padded_$i.err
timeout -k 10 300 python bench.py --model roberta --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline --graph > gpurun_out/r6b/c3_graph_$i.json 2> gpurun_out/r6b/c3_graph_$i.err
done
grep -o '"value": [0-9.]*' gpurun_out/r6b/c3_*.json
echo finished
