#!/bin/bash
# k_mscan measurement switches on the final kernel (AGH_MSCAN_DBG is read when the query is built)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r06_perf_c5_dbg.log
for dbg in 0 8 0 8 1 2 4 16; do
  echo -n "AGH_MSCAN_DBG=$dbg " >> gpurun_out/r06_perf_c5_dbg.log
  AGH_MSCAN_DBG=$dbg python scripts/perf_c5_quick.py 2>&1 | grep "^c5" >> gpurun_out/r06_perf_c5_dbg.log
done
cat gpurun_out/r06_perf_c5_dbg.log
