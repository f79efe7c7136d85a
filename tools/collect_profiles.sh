#!/bin/bash
# Everything profiles/ holds for one round, in GPU calls of <= 15 minutes:  tools/collect_profiles.sh r05 lines|stats|pmc
# (writes gpurun_out/<tag>/; copy what is to be judged into profiles/).  The fp32 headline is the library default: split
# products, padding-free execution, pre-split operand images (round 5); `--f32-pipe` is the fp32 MFMA pipe, `--padded` the padded
# layout (in-kernel split), MTVAF_F32_PLANES=0 the in-kernel split on packed rows.
export TMPDIR=/tmp
T=${1:-r05}
PHASE=${2:-lines}
R=$PWD
O=$R/gpurun_out/$T
mkdir -p $O
S="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary"   # 7 model steps per process (optimizer included since round 5)
stats() {  # <name> <env assignment> <bench args...>
  local name=$1 envv=$2; shift 2
  ( export $envv; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 bench.py $S "$@" > $O/prof_$name.log 2>&1 )
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | tail -1) $O/${T}_${name}_kernel_stats.csv
  python tools/prof_summary.py $O/${T}_${name}_kernel_stats.csv 7 60 > $O/${T}_${name}_kernel_stats_per_step.txt
  rm -rf $O/prof_$name
}
if [ $PHASE = lines ]; then
python bench.py --steps 20 --warmup 5 > $O/${T}_bench_line.json 2> $O/bench_fp32.err
cp bench_detail.json $O/${T}_bench_detail.json
python bench.py --steps 20 --warmup 5 --f32-pipe --no-cpu-baseline --no-secondary > $O/${T}_bench_line_fp32_pipe.json 2> $O/bench_pipe.err
python bench.py --steps 20 --warmup 5 --full-length --no-cpu-baseline --no-secondary > $O/${T}_bench_line_full_length.json 2> $O/bench_full.err
python bench.py --steps 20 --warmup 5 --padded --no-cpu-baseline --no-secondary > $O/${T}_bench_line_padded.json 2> $O/bench_padded.err
python bench.py --steps 20 --warmup 5 --dtype bf16 --model roberta --no-cpu-baseline --no-secondary > $O/${T}_bench_line_bf16_c3.json 2> $O/bench_c3.err
python bench.py --steps 20 --warmup 5 --dtype bf16 --batch 64 --no-cpu-baseline --no-secondary > $O/${T}_bench_line_bf16_c4.json 2> $O/bench_c4.err
python bench.py --steps 5 --warmup 4 --dtype bf16 --batch 128 --seq 512 --no-cpu-baseline --no-secondary > $O/${T}_bench_line_bf16_c5.json 2> $O/bench_c5.err
python bench.py --steps 5 --warmup 4 --batch 128 --seq 512 --no-cpu-baseline --no-secondary > $O/${T}_bench_line_fp32_c5.json 2> $O/bench_c5f.err
python tools/f32x3_bench.py > $O/${T}_f32x3_microbench.txt 2>/dev/null
python tools/f32x3_bench.py 2432 >> $O/${T}_f32x3_microbench.txt 2>/dev/null
python tools/x3_trace.py 4096 3072 768 5 > $O/${T}_x3_trace.txt 2>/dev/null
python tools/x3_trace.py 4096 768 768 6 >> $O/${T}_x3_trace.txt 2>/dev/null
echo "== weight gradient (KM x KM), 768 x 3072 x 2048 ==" >> $O/${T}_x3_trace.txt
python tools/x3_trace.py 768 3072 2048 5 1 1 2>/dev/null | tail -4 >> $O/${T}_x3_trace.txt
echo "== dX (KC x KM), 4096 x 3072 x 768 ==" >> $O/${T}_x3_trace.txt
python tools/x3_trace.py 4096 3072 768 5 0 1 2>/dev/null | tail -4 >> $O/${T}_x3_trace.txt
# sustained bf16 MFMA rate of the whole chip (hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/micro/mfma_bf16_rate.hip)
[ -x tools/micro/mfma_bf16_rate ] && tools/micro/mfma_bf16_rate > $O/${T}_mfma_bf16_rate.txt 2>&1
fi
if [ $PHASE = stats ]; then
stats fp32 MTVAF_DW_STREAM=1
stats fp32_serial MTVAF_DW_STREAM=0
stats fp32_insplit_serial "MTVAF_DW_STREAM=0 MTVAF_F32_PLANES=0"
stats fp32_padded_serial MTVAF_DW_STREAM=0 --padded
stats fp32_pipe_serial MTVAF_DW_STREAM=0 --f32-pipe
stats bf16_c3_serial MTVAF_DW_STREAM=0 --dtype bf16 --model roberta
stats bf16_c4_serial MTVAF_DW_STREAM=0 --dtype bf16 --batch 64
fi
if [ $PHASE = pmc ]; then
bash tools/pmc_passes.sh $T/pmc_fp32 > /dev/null
bash tools/pmc_passes.sh $T/pmc_fp32_padded --padded > /dev/null
python tools/pmc_to_json.py $O/pmc_fp32 $O $T pmc_gemm.json > /dev/null
python tools/pmc_to_json.py $O/pmc_fp32_padded $O $T pmc_gemm_padded.json > /dev/null
rm -rf $O/pmc_fp32 $O/pmc_fp32_padded
fi
date; ls $O
