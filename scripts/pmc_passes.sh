#!/bin/bash
# PMC passes over the C2 workload (one rocprofv3 run per counter group; no trace domains
# besides --kernel-trace).  Usage: scripts/pmc_passes.sh <outdir-under-gpurun_out> [gib] [k] [lean|numbered|full]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
GIB=${2:-4}
K=${3:-2}
MODE=${4:-lean}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/sq1 -o p -- python $GRAFT_REPO_ROOT/scripts/prof_k2.py $GIB $K $MODE > $OUT.sq1.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/tcc -o p -- python $GRAFT_REPO_ROOT/scripts/prof_k2.py $GIB $K $MODE > $OUT.tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python $GRAFT_REPO_ROOT/scripts/prof_k2.py $GIB $K $MODE > $OUT.sq2.log 2>&1
# how the HBM reads are made up: 32-byte vs 64-byte read requests of the L2 (settles what FETCH_SIZE x 2 means
# for a kernel that reads 64-byte segments)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum \
  --kernel-trace --output-format csv -d $OUT/tcc2 -o p -- python $GRAFT_REPO_ROOT/scripts/prof_k2.py $GIB $K $MODE > $OUT.tcc2.log 2>&1
find $OUT -name "*.csv" | head -20
tail -2 $OUT.sq1.log
