#!/bin/bash
# A/B of the neighbour-byte filter in front of level 3 of the one-pass -f kernel (AGH_MS_NBF, agh_mscan.hip) --
# written at the end of round 4 without GPU time left, NOT measured yet.  Build the variant on the CPU box first:
#   make -C agrep_amd/csrc -j10 VARIANT=nbf VARFLAGS="-DAGH_MS_NBF=1"
# then, on the GPU box (one gpurun call):
#   bash scripts/ab_c5_nbf.sh
# 1. parity of the variant: the one-pass tests and the full-size C5 tests; 2. the -f sets on 4 GiB under both libraries.
set -u
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/agrep_amd/libagrep_hip_nbf.so
[ -f $V ] || { echo "build the variant first (see the header of this script)"; exit 1; }
export AGH_REQUIRE_GPU=1
(AGH_LIB_PATH=$V timeout 200 python -m pytest tests/test_gpu_multi.py -q -x -k "one_pass" -o timeout=150 2>&1 | tail -4) < /dev/null
(AGH_LIB_PATH=$V timeout 200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c5" -o timeout=150 2>&1 | tail -4) < /dev/null
echo "== shipped build (2^12 rows for a like-for-like table size)"
(AGH_MSCAN_RB=12 timeout 120 python scripts/perf_c5_r4.py 4 5 2>&1 | grep -v amdgpu.ids) < /dev/null
echo "== AGH_MS_NBF=1"
(AGH_LIB_PATH=$V timeout 120 python scripts/perf_c5_r4.py 4 5 2>&1 | grep -v amdgpu.ids) < /dev/null
