cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m
timeout -k 10 300 python -m pytest tests/test_unpad_gpu.py -m gpu -x -q -k "ordered_packing" > gpurun_out/r6m/tests.log 2>&1; tail -15 gpurun_out/r6m/tests.log
