#!/bin/bash
# Everything DESIGN.md quotes, in one GPU call:  scripts/evidence_run.sh r02
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${TAG}_bench_line.json
scripts/profile_configs.sh $TAG > gpurun_out/${TAG}_profile_configs.log 2>&1
scripts/pmc_passes.sh pmc_${TAG}_fullc 2 2 fullc > gpurun_out/${TAG}_pmc_fullc.log 2>&1
AGH_FUSED_MIN_MB=0 scripts/pmc_passes.sh pmc_${TAG}_lean 8 2 lean > gpurun_out/${TAG}_pmc_lean.log 2>&1
scripts/pmc_passes.sh pmc_${TAG}_multi 2 0 multi > gpurun_out/${TAG}_pmc_multi.log 2>&1
python scripts/pmc_summary.py pmc_${TAG}_multi k_sweep_multi 2147483648 gpurun_out/${TAG}_pmc_sweep_multi.json "prof_k2.py 2 GiB, 1024 exact patterns (4..12 B), count-only"
python scripts/pmc_summary.py pmc_${TAG}_fullc k_fullscan 2147483648 gpurun_out/${TAG}_pmc_fullscan.json "prof_k2.py 2 GiB, k=2, count-only full scan (AGH_FORCE_FULLSCAN|AGH_COUNT)"
python scripts/pmc_summary.py pmc_${TAG}_lean "k_sweep_fused" 8589934592 gpurun_out/${TAG}_pmc_sweep.json "prof_k2.py 8 GiB, k=2, count-only (the headline kernel: sweep + verify fused, one segment)"
for g in 4 8 16 64; do AGH_AB_MODES=0,1,0,1 python scripts/ab_fused.py $g 10 2>&1 | grep -v "^/opt\|n="; done > gpurun_out/${TAG}_ab_fused.log
python scripts/perf_multi.py 4 2>&1 | tail -7 > gpurun_out/${TAG}_perf_multi.log
python scripts/perf_multi_strided.py 2>&1 | tail -4 > gpurun_out/${TAG}_perf_multi_strided.log
python scripts/perf_short.py 2>&1 | tail -9 > gpurun_out/${TAG}_perf_piece_engine.log
python scripts/perf_fullscan.py 4 2>&1 | tail -12 > gpurun_out/${TAG}_perf_fallback_engines.log
python scripts/c5_files.py 8 512 2>&1 | tail -3 > gpurun_out/${TAG}_c5_files.log
ls gpurun_out | grep "^${TAG}_" | head -40
