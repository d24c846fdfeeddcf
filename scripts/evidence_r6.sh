#!/bin/bash
# Round-6 evidence in one GPU call, on the final code: the bench line (headline + c2_records / c3 (+ pinned m = 29) / c5 /
# c5_as_worded / engines blocks + cpu_baseline over all shards on both of the reference's engine paths + both traffic
# passes), its rocprofv3 kernel-trace summary, PMC passes of k_mtile, k_mscan and the three-stream full scan, the
# dense-set A/B, the GPU suite.  Everything lands in gpurun_out/r06_*.
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python bench.py 2>gpurun_out/r06_bench.err | tail -1 > gpurun_out/r06_bench_line.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06_bench -o bench -- \
    python $R/bench.py --steps 20 --warmup 2 --no-traffic --no-cpu-baseline --config-steps 5 > $R/gpurun_out/r06_bench_line_under_rocprof.json 2>/dev/null)
f=$(find gpurun_out/prof_r06_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r06_bench_kernel_stats.csv
rm -rf gpurun_out/prof_r06_bench
bash scripts/pmc_passes.sh r06_pmc_mtile 1 1 multi > /dev/null 2>&1
python scripts/pmc_summary.py r06_pmc_mtile "k_mtile<" 1073741824 gpurun_out/r06_pmc_mtile.json \
    "prof_k2.py 1 GiB, 1024 patterns (4..12 B), k=1, count-only: the tile kernel (final)" | cut -c1-700
rm -rf gpurun_out/r06_pmc_mtile gpurun_out/r06_pmc_mtile.*.log
bash scripts/pmc_passes.sh r06_pmc_mscan 4 1 multik > /dev/null 2>&1
python scripts/pmc_summary.py r06_pmc_mscan "k_mscan<" 4294967296 gpurun_out/r06_pmc_mscan.json \
    "prof_k2.py 4 GiB, 1024 patterns (8..12 B), k=1, count-only: the one-pass kernel (two supertiles in flight, one entry load)" | cut -c1-700
rm -rf gpurun_out/r06_pmc_mscan gpurun_out/r06_pmc_mscan.*.log
bash scripts/pmc_passes.sh r06_pmc_word 4 2 word > /dev/null 2>&1
python scripts/pmc_summary.py r06_pmc_word "k_fullscan_fast<" 4294967296 gpurun_out/r06_pmc_fullscan_fast3.json \
    "prof_k2.py 4 GiB, 'matching' k=2, forced full scan, count-only: three text streams per lane" | cut -c1-700
rm -rf gpurun_out/r06_pmc_word gpurun_out/r06_pmc_word.*.log
timeout 300 python scripts/perf_c5_worded_r6.py 4 2>&1 | grep "^c5" > gpurun_out/r06_perf_c5_worded_final.log
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06_pytest_gpu.log
cut -c1-400 gpurun_out/r06_bench_line.json
tail -3 gpurun_out/r06_bench.err
head -12 gpurun_out/r06_bench_kernel_stats.csv | cut -c1-160
grep -v "SHARE\|DBG" gpurun_out/r06_perf_c5_worded_final.log | tail -5
cat gpurun_out/r06_pytest_gpu.log
