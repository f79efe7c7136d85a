#!/bin/bash
set -e
python -m pytest tests/test_optim_gpu.py tests/test_parallel.py tests/test_unpad_gpu.py -m gpu -x -q 2>&1 | tail -3
