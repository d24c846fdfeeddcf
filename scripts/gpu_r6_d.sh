#!/bin/bash
# k_mscan: two supertiles in flight against round 5's one (libagrep_hip_old.so), interleaved on one box
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r06_ab_mscan_prefetch.log
for rnd in 1 2 3; do
  AGH_LIB_PATH=$PWD/agrep_amd/libagrep_hip_old.so python scripts/perf_c5_quick.py 2>&1 | grep "^c5" >> gpurun_out/r06_ab_mscan_prefetch.log
  python scripts/perf_c5_quick.py 2>&1 | grep "^c5" >> gpurun_out/r06_ab_mscan_prefetch.log
done
cat gpurun_out/r06_ab_mscan_prefetch.log
timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "one_pass" 2>&1 | tail -3
