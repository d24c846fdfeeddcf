#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "record_walk" 2>&1 | tail -3
timeout 600 python scripts/perf_c5_worded_r6.py 4 2>&1 | grep "^c5" | grep -v "SHARE" > gpurun_out/r06_ab_mtile_v6.log
cat gpurun_out/r06_ab_mtile_v6.log
