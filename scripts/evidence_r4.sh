#!/bin/bash
# Round-4 evidence in one GPU call, on the final code: the bench line (headline + c2_records / c3 / c5 blocks +
# cpu_baseline over all shards + both traffic passes), its rocprofv3 kernel-trace summary, the PMC passes of the
# one-pass -f kernel, the -f sets on resident text, and the C5 file job.  Everything lands in gpurun_out/r04_*.
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python bench.py 2>gpurun_out/r04_bench.err | tail -1 > gpurun_out/r04_bench_line.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_bench -o bench -- \
    python $R/bench.py --steps 20 --warmup 2 --no-traffic --no-cpu-baseline --config-steps 5 > $R/gpurun_out/r04_bench_line_under_rocprof.json 2>/dev/null)
f=$(find gpurun_out/prof_r04_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r04_bench_kernel_stats.csv
rm -rf gpurun_out/prof_r04_bench
bash scripts/pmc_passes.sh r04_pmc_mscan_final 2 1 multik > /dev/null 2>&1
python scripts/pmc_summary.py r04_pmc_mscan_final "k_mscan<" 2147483648 gpurun_out/r04_pmc_mscan.json \
    "prof_k2.py 2 GiB, 1024 patterns (8..12 B), k=1, count-only: the one-pass kernel (sweep + verification)" | cut -c1-600
rm -rf gpurun_out/r04_pmc_mscan_final
timeout 200 python scripts/perf_c5_r4.py 4 5 2>&1 | grep -v "^/opt" > gpurun_out/r04_perf_c5.log
timeout 300 python scripts/c5_files.py 32 1024 1 2>&1 | grep -v "^/opt" > gpurun_out/r04_c5_files.log
cut -c1-300 gpurun_out/r04_bench_line.json
tail -3 gpurun_out/r04_bench.err
cat gpurun_out/r04_c5_files.log
