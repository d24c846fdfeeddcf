#!/bin/bash
# Two PMC passes (SQ issue counters, FETCH_SIZE) over the table engine's two-streams kernel: 'approx#match', k from argv,
# 2 GiB, count-only.  One rocprofv3 run per counter group, --kernel-trace only.  Usage: scripts/pmc_table_r4.sh <outdir> [k]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_table}
K=${2:-0}
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/sq1 -o p -- python $GRAFT_REPO_ROOT/scripts/prof_k2.py 2 $K table > $OUT.sq1.log 2>&1 < /dev/null
timeout 150 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/tcc -o p -- python $GRAFT_REPO_ROOT/scripts/prof_k2.py 2 $K table > $OUT.tcc.log 2>&1 < /dev/null
find $OUT -name "*.csv" < /dev/null | head -20
