#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_multi.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_c_pytest_multi.log
cat gpurun_out/r06_c_pytest_multi.log
