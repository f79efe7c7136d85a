cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6q
for rep in 1 2 3; do for k in 0 1; do
  MTVAF_ATTN_SPLIT=$k timeout -k 10 300 python bench.py --steps 30 --warmup 8 --full-length --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r6q/full_s${k}_$rep.json 2> gpurun_out/r6q/full_s${k}_$rep.err
done; done
grep -o '"value": [0-9.]*' gpurun_out/r6q/*.json
