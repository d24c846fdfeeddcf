#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./scripts/micro_lds_unaligned.bin > gpurun_out/r06_micro_lds_unaligned.log 2>&1
cat gpurun_out/r06_micro_lds_unaligned.log
for rnd in 1 2 3; do python scripts/perf_c5_quick.py 2>&1 | grep "^c5"; done | tee gpurun_out/r06_perf_c5_one_load.log
timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "one_pass or c5" 2>&1 | tail -3
