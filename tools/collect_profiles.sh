#!/bin/bash
# Everything profiles/ holds for one round, in one GPU call:  tools/collect_profiles.sh r03   (writes gpurun_out/<tag>/)
# Bench lines, rocprofv3 kernel stats (default = two streams, and serialised = one kernel at a time), PMC passes.
export TMPDIR=/tmp
T=${1:-r03}
PHASE=${2:-all}   # lines | stats | pmc | all  (one GPU call holds 20 minutes: the round-3 set is collected in three)
R=$PWD
O=$R/gpurun_out/$T
mkdir -p $O
S="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-optimizer --no-secondary"   # 7 model steps per process
stats() {  # <name> <env> <bench args...>
  local name=$1 envv=$2; shift 2
  env $envv TMPDIR=/tmp true
  ( export $envv; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 bench.py $S "$@" > $O/prof_$name.log 2>&1 )
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | tail -1) $O/${T}_${name}_kernel_stats.csv
  python tools/prof_summary.py $O/${T}_${name}_kernel_stats.csv 7 60 > $O/${T}_${name}_kernel_stats_per_step.txt
  rm -rf $O/prof_$name
}
if [ $PHASE = all ] || [ $PHASE = lines ]; then
echo "== bench lines"; date
python bench.py --steps 20 --warmup 5 > $O/${T}_bench_line.json 2> $O/bench_fp32.err
python bench.py --steps 20 --warmup 5 --dtype bf16 --model roberta --no-cpu-baseline > $O/${T}_bench_line_bf16_c3.json 2> $O/bench_c3.err
python bench.py --steps 20 --warmup 5 --dtype bf16 --batch 64 --no-cpu-baseline > $O/${T}_bench_line_bf16_c4.json 2> $O/bench_c4.err
python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline > $O/${T}_bench_line_bf16_c2shape.json 2> $O/bench_c2b.err
python bench.py --steps 5 --warmup 4 --dtype bf16 --batch 128 --seq 512 --no-cpu-baseline > $O/${T}_bench_line_bf16_c5.json 2> $O/bench_c5.err
python bench.py --steps 20 --warmup 5 --unpad --no-cpu-baseline > $O/${T}_bench_line_unpad.json 2> $O/bench_unpad.err
python bench.py --steps 20 --warmup 5 --unpad --dtype bf16 --batch 64 --no-cpu-baseline --no-roofline > $O/${T}_bench_line_unpad_bf16_c4.json 2> $O/bench_unpad_c4.err
MTVAF_F32_SPLIT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/${T}_bench_line_fp32_split.json 2> $O/bench_split.err
MTVAF_F32_SPLIT=1 python bench.py --steps 20 --warmup 5 --unpad --no-cpu-baseline --no-secondary > $O/${T}_bench_line_fp32_split_unpad.json 2> $O/bench_split_unpad.err
python tools/f32x3_bench.py > $O/${T}_f32x3_microbench.txt 2>/dev/null
fi
if [ $PHASE = all ] || [ $PHASE = stats ]; then
echo "== kernel stats"; date
stats fp32 MTVAF_DW_STREAM=1
stats fp32_serial MTVAF_DW_STREAM=0
stats fp32_unpad_serial MTVAF_DW_STREAM=0 --unpad
stats fp32_split_serial "MTVAF_DW_STREAM=0 MTVAF_F32_SPLIT=1"
stats bf16_c3_serial MTVAF_DW_STREAM=0 --dtype bf16 --model roberta
stats bf16_c4_serial MTVAF_DW_STREAM=0 --dtype bf16 --batch 64
stats bf16_c4 MTVAF_DW_STREAM=1 --dtype bf16 --batch 64
S="--steps 3 --warmup 4 --no-cpu-baseline --no-roofline --no-optimizer --no-secondary"
stats bf16_c5_serial MTVAF_DW_STREAM=0 --dtype bf16 --batch 128 --seq 512
fi
if [ $PHASE = all ] || [ $PHASE = pmc ]; then
echo "== pmc"; date
bash tools/pmc_passes.sh $T/pmc_fp32 > /dev/null
bash tools/pmc_passes.sh $T/pmc_bf16_c4 --dtype bf16 --batch 64 > /dev/null
bash tools/pmc_passes.sh $T/pmc_bf16_c3 --dtype bf16 --model roberta > /dev/null
bash tools/pmc_passes.sh $T/pmc_bf16_c5 --dtype bf16 --batch 128 --seq 512 > /dev/null
python tools/pmc_to_json.py $O/pmc_fp32 $O $T pmc_gemm.json > /dev/null
python tools/pmc_to_json.py $O/pmc_bf16_c4 $O $T pmc_gemm_bf16_b64.json > /dev/null
python tools/pmc_to_json.py $O/pmc_bf16_c3 $O $T pmc_gemm_bf16_b32.json > /dev/null
python tools/pmc_to_json.py $O/pmc_bf16_c5 $O $T pmc_gemm_bf16_b128.json > /dev/null
rm -rf $O/pmc_fp32 $O/pmc_bf16_c4 $O/pmc_bf16_c3 $O/pmc_bf16_c5
# the split-fp32 GEMM alone (FFN-1 forward + its weight gradient): MFMA busy, waits, co-execution, LDS
echo "rocprofv3 --kernel-trace --pmc <group> -- python3 tools/f32x3_pmc.py (one MI355X; per-dispatch means over 10 launches)" > $O/${T}_f32x3_pmc.txt
for g in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/x3pmc -- python3 tools/f32x3_pmc.py > /dev/null 2>&1
  python tools/pmc_summary.py $(find $O/x3pmc -name "*counter_collection.csv" | tail -1) f32x3 >> $O/${T}_f32x3_pmc.txt
  rm -rf $O/x3pmc
done
fi
date; ls -la $O
