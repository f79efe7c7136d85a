#!/bin/bash
# PMC passes for the bench (one counter group per run; rocprofv3 --pmc must not be mixed with sys traces).
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> [extra bench.py args ...]     e.g.  tools/pmc_passes.sh pmc_bf16 --dtype bf16
export TMPDIR=/tmp
# per-kernel counters are only meaningful when kernels run one at a time: serialise the weight-gradient stream
export MTVAF_DW_STREAM=0
R=$PWD
OUT=$R/gpurun_out/${1:-pmc}
shift
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary $*"
MOPS="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16"  # (the split-fp32 kernels run the bf16 pipe)

rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY $MOPS SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p3 -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/p4 -- python3 $ARGS > /dev/null 2>&1
find $OUT -name "*counter_collection.csv" | head -8
