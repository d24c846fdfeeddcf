#!/bin/bash
# Round 5: the same-box A/B of the experimental variants of the one-pass -f kernel (agh_mscan.hip) the round-4 review
# asked for: shipped, AGH_MS_NBF=1, AGH_MS_NBF=2 (2^12 rows), AGH_MS_L3PIPE=1; config 5 (1024 x 8..12 B, k = 1) on
# 4 GiB, device time of the whole count-only scan, median of 5, three interleaved rounds.  Build the variants first:
# (the variants were built from the round-4 tree with VARFLAGS -DAGH_MS_NBF=1 / =2 / -DAGH_MS_L3PIPE=1; the code is gone: none was faster)
set -u
cd $GRAFT_REPO_ROOT
export AGH_REQUIRE_GPU=1
L=$GRAFT_REPO_ROOT/agrep_amd
echo "== parity of the never-run variant (l3pipe): one-pass tests"
(AGH_LIB_PATH=$L/libagrep_hip_l3pipe.so timeout 300 python -m pytest tests/test_gpu_multi.py -q -x -k "one_pass" -o timeout=200 2>&1 | tail -2) < /dev/null
for i in 1 2 3; do
  echo "== round $i"
  (timeout 60 python scripts/perf_c5_quick.py 2>&1 | grep "^c5") < /dev/null
  (AGH_MSCAN_RB=12 timeout 60 python scripts/perf_c5_quick.py 2>&1 | grep "^c5" | sed 's/$/ [rb=12]/') < /dev/null
  for v in nbf nbf2 l3pipe; do
    (AGH_LIB_PATH=$L/libagrep_hip_$v.so timeout 60 python scripts/perf_c5_quick.py 2>&1 | grep "^c5") < /dev/null
  done
done
