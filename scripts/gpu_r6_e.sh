#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r06_e_pytest_parity.log
timeout 600 python scripts/perf_fullscan_r6.py 2>&1 | grep "^fullscan" | tee gpurun_out/r06_ab_fullscan_streams_final.log
