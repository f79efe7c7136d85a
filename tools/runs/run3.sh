set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_graph_gpu.py tests/test_properties_gpu.py tests/test_frontend.py -m gpu -q -s > $O/gputest3.log 2>&1; echo "pytest rc=$?"
tail -4 $O/gputest3.log; grep "bf16 cache build" $O/gputest3.log
S="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary"
for name in unpad unpad_serial; do
  if [ $name = unpad_serial ]; then export MTVAF_DW_STREAM=0; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 bench.py $S > $O/prof_$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | tail -1) $O/r05_fp32_${name}_kernel_stats.csv
  python tools/prof_summary.py $O/r05_fp32_${name}_kernel_stats.csv 7 60 > $O/r05_fp32_${name}_kernel_stats_per_step.txt
  rm -rf $O/prof_$name
done
head -40 $O/r05_fp32_unpad_serial_kernel_stats_per_step.txt
