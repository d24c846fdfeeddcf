#!/bin/bash
# A/B of the neighbour-byte filters of the one-pass -f kernel (AGH_MS_NBF, agh_mscan.hip): 1 = four neighbour bytes
# tested at the head of level 3, 2 = the fifth byte tested in level 2 against a mask table in LDS.  Written at the
# end of round 4 with seconds of GPU time left: both are parity-green (profiles/r04_nbf*_quick.log) but were timed
# on different boxes -- this script is the same-box A/B.  Build the variants on the CPU box first:
#   make -C agrep_amd/csrc -j10 VARIANT=nbf VARFLAGS="-DAGH_MS_NBF=1"
#   make -C agrep_amd/csrc -j10 VARIANT=nbf2 VARFLAGS="-DAGH_MS_NBF=2"
#   make -C agrep_amd/csrc -j10 VARIANT=l3pipe VARFLAGS="-DAGH_MS_L3PIPE=1"     (level 3 in two halves, one batch in flight:
#                                                                               written after the last GPU second, never run)
# then, on the GPU box (one gpurun call):
#   bash scripts/ab_c5_nbf.sh
# 1. parity of the variant: the one-pass tests and the full-size C5 tests; 2. the -f sets on 4 GiB under both libraries.
set -u
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/agrep_amd/libagrep_hip_nbf.so
[ -f $V ] || { echo "build the variants first (see the header of this script)"; exit 1; }
export AGH_REQUIRE_GPU=1
(AGH_LIB_PATH=$V timeout 200 python -m pytest tests/test_gpu_multi.py -q -x -k "one_pass" -o timeout=150 2>&1 | tail -4) < /dev/null
(AGH_LIB_PATH=$V timeout 200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c5" -o timeout=150 2>&1 | tail -4) < /dev/null
echo "== shipped build (2^12 rows for a like-for-like table size)"
(AGH_MSCAN_RB=12 timeout 120 python scripts/perf_c5_r4.py 4 5 2>&1 | grep -v amdgpu.ids) < /dev/null
echo "== AGH_MS_NBF=1"
(AGH_LIB_PATH=$V timeout 120 python scripts/perf_c5_r4.py 4 5 2>&1 | grep -v amdgpu.ids) < /dev/null
V3=$GRAFT_REPO_ROOT/agrep_amd/libagrep_hip_l3pipe.so
if [ -f $V3 ]; then
  echo "== AGH_MS_L3PIPE=1"
  (AGH_LIB_PATH=$V3 timeout 200 python -m pytest tests/test_gpu_multi.py -q -x -k "one_pass" -o timeout=150 2>&1 | tail -4) < /dev/null
  (AGH_LIB_PATH=$V3 timeout 200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c5" -o timeout=150 2>&1 | tail -4) < /dev/null
  for i in 1 2 3; do
    (timeout 30 python scripts/perf_c5_quick.py 2>&1 | grep "^c5") < /dev/null
    (AGH_LIB_PATH=$V3 timeout 30 python scripts/perf_c5_quick.py 2>&1 | grep "^c5") < /dev/null
  done
fi
V2=$GRAFT_REPO_ROOT/agrep_amd/libagrep_hip_nbf2.so
if [ -f $V2 ]; then
  echo "== AGH_MS_NBF=2"
  (AGH_LIB_PATH=$V2 timeout 200 python -m pytest tests/test_gpu_multi.py -q -x -k "one_pass" -o timeout=150 2>&1 | tail -4) < /dev/null
  (AGH_LIB_PATH=$V2 timeout 120 python scripts/perf_c5_r4.py 4 5 2>&1 | grep -v amdgpu.ids) < /dev/null
  for i in 1 2 3; do
    (AGH_MSCAN_RB=12 timeout 30 python scripts/perf_c5_quick.py 2>&1 | grep "^c5") < /dev/null
    (AGH_LIB_PATH=$V timeout 30 python scripts/perf_c5_quick.py 2>&1 | grep "^c5") < /dev/null
    (AGH_LIB_PATH=$V2 timeout 30 python scripts/perf_c5_quick.py 2>&1 | grep "^c5") < /dev/null
  done
fi
