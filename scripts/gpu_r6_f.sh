#!/bin/bash
# the whole GPU suite on the current code
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r06_pytest_gpu.log
