set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
timeout -k 10 400 python tools/f32p_bench.py 4096 2432 > $O/f32p_bench.txt 2>&1; echo "rc=$?"
cat $O/f32p_bench.txt | cut -c1-400
