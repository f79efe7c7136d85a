#!/bin/bash
# tile walk G = 4 as the default: the pre-split kernels' tests, then the counter passes of the default step
set -e
O=gpurun_out/r05e
mkdir -p $O
python -m pytest tests/test_ops_gpu.py tests/test_unpad_gpu.py -m gpu -x -q -k "presplit or plane" 2>&1 | tail -3
bash tools/pmc_passes.sh r05e/pmc_fp32 > /dev/null
python tools/pmc_to_json.py $O/pmc_fp32 $O r05e pmc_gemm.json > $O/pmc_to_json.log 2>&1 || { tail $O/pmc_to_json.log; exit 1; }
rm -rf $O/pmc_fp32
ls $O
