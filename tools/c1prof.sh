export TMPDIR=/tmp
O=$PWD/gpurun_out/c1prof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --batch 4 --seq 64 --aux 3 > $O/log.txt 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | tail -1) $O/c1_kernel_stats.csv
python tools/prof_summary.py $O/c1_kernel_stats.csv 7 70 > $O/c1_per_step.txt
rm -rf $O/prof; head -70 $O/c1_per_step.txt
