cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
for rep in 1 2; do for k in 0 1; do
  MTVAF_P16_PERSIST=$k timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/r6h/c2_p${k}_$rep.json 2> gpurun_out/r6h/c2_p${k}_$rep.err
done; done
for k in 0 1; do
  MTVAF_P16_PERSIST=$k timeout -k 10 300 python bench.py --steps 20 --warmup 5 --full-length --no-cpu-baseline --no-secondary > gpurun_out/r6h/full_p$k.json 2> gpurun_out/r6h/full_p$k.err
  MTVAF_P16_PERSIST=$k timeout -k 10 300 python bench.py --steps 5 --warmup 3 --batch 128 --seq 512 --no-cpu-baseline --no-secondary > gpurun_out/r6h/c5_p$k.json 2> gpurun_out/r6h/c5_p$k.err
done
for f in gpurun_out/r6h/*.json; do echo "$f $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"kernel": "[^"]*"' $f | head -1) $(grep -o '"avg_launch_us": [0-9.]*' $f | head -1) $(grep -o '"frac": [0-9.]*' $f | head -1) $(grep -o '"all_gemm_frac": [0-9.]*' $f | head -1)"; done
