#!/bin/bash
# After scripts/evidence_r3.sh the fused kernel's workgroup count (3 per CU, 2 for H >= 8) and its size
# threshold (4 GiB) changed: the lines that depend on them again, on the final code, in one GPU call.
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -q -m gpu > gpurun_out/r03_pytest_gpu_fused_defaults.log 2>&1; echo pytest rc $?)
python scripts/ab_final_r3.py 64 10 2>&1 | grep -v "^/opt" > gpurun_out/r03_ab_headline_final.log
python bench.py 2>gpurun_out/r03_bench.err | tail -1 > gpurun_out/r03_bench_line.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_bench -o bench -- \
    python $R/bench.py --steps 10 --warmup 2 --no-traffic --no-cpu-baseline > $R/gpurun_out/r03_bench_line_under_rocprof.json 2>/dev/null)
f=$(find gpurun_out/prof_r03_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_bench_kernel_stats.csv
python scripts/config_runs.py c3 2>&1 | grep -v "^/opt" > gpurun_out/r03_c3.log
scripts/pmc_r3_final.sh > gpurun_out/r03_pmc_final.log 2>&1
tail -3 gpurun_out/r03_pytest_gpu_fused_defaults.log
cut -c1-400 gpurun_out/r03_bench_line.json
