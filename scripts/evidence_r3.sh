#!/bin/bash
# Everything DESIGN.md quotes for round 3, in one GPU call:  scripts/evidence_r3.sh
# (the PMC summaries come from scripts/pmc_r3.sh, a call of its own)
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python bench.py 2>gpurun_out/r03_bench.err | tail -1 > gpurun_out/r03_bench_line.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_bench -o bench -- \
    python $R/bench.py --steps 10 --warmup 2 --no-traffic --no-cpu-baseline > $R/gpurun_out/r03_bench_line_under_rocprof.json 2>/dev/null)
f=$(find gpurun_out/prof_r03_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_bench_kernel_stats.csv
# the headline A/B (fused vs two kernels, ticket tail, H = 2) on the final build, 64 and 8 GiB
python scripts/ab_final_r3.py 64 10 2>&1 | grep -v "^/opt" > gpurun_out/r03_ab_headline_final.log
python scripts/config_runs.py c3 2>&1 | grep -v "^/opt" > gpurun_out/r03_c3.log
python scripts/perf_fullscan.py 4 2>&1 | grep -v "^/opt" > gpurun_out/r03_perf_fallback_engines.log
AGH_FS_FAST=0 python scripts/perf_fullscan.py 4 2>&1 | grep -E "tablescan" > gpurun_out/r03_perf_tablescan_exact_kernel.log
python scripts/perf_short.py 2>&1 | grep -v "^/opt" > gpurun_out/r03_perf_piece_engine.log
python scripts/perf_multi_r3.py 4 2>&1 | grep -v "^/opt" | cut -c1-260 > gpurun_out/r03_perf_multi.log
python scripts/perf_fullscan_r3.py 4 2>&1 | grep -v "^/opt" > gpurun_out/r03_perf_fullscan.log
ls gpurun_out | grep "^r03_" | head -60
