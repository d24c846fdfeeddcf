#!/bin/bash
# Round-5 evidence in one GPU call, on the final code: the bench line (headline + c2_records / c3 / c5 / c5_as_worded
# blocks + cpu_baseline over all shards WITH the sha256 of the matched records + both traffic passes), its rocprofv3
# kernel-trace summary, the PMC passes of the numbered verify kernel and of the record walk, the record paths and
# config 5 as worded on resident text, file -> records, and the start-up timeline.  Everything lands in gpurun_out/r05_*.
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python bench.py 2>gpurun_out/r05_bench.err | tail -1 > gpurun_out/r05_bench_line.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05_bench -o bench -- \
    python $R/bench.py --steps 20 --warmup 2 --no-traffic --no-cpu-baseline --config-steps 5 > $R/gpurun_out/r05_bench_line_under_rocprof.json 2>/dev/null)
f=$(find gpurun_out/prof_r05_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r05_bench_kernel_stats.csv
rm -rf gpurun_out/prof_r05_bench
# PMC: the numbered pipeline of the headline pattern on 4 GiB (k_sweep<H=2, census> + k_verify numbered)
bash scripts/pmc_passes.sh r05_pmc_numbered 4 2 numbered > /dev/null 2>&1
python scripts/pmc_summary.py r05_pmc_numbered "k_verify<" 4294967296 gpurun_out/r05_pmc_verify_numbered.json \
    "prof_k2.py 4 GiB, m=16 k=2, numbered scan: the verifying kernel (tight windows, one window per occurrence)" | cut -c1-400
python scripts/pmc_summary.py r05_pmc_numbered "k_sweep<" 4294967296 gpurun_out/r05_pmc_sweep_numbered.json \
    "prof_k2.py 4 GiB, m=16 k=2, numbered scan: the census sweep" | cut -c1-400
rm -rf gpurun_out/r05_pmc_numbered gpurun_out/r05_pmc_numbered.*.log
bash scripts/pmc_passes.sh r05_pmc_mwalk 1 1 multi > /dev/null 2>&1
python scripts/pmc_summary.py r05_pmc_mwalk "k_mwalk<" 1073741824 gpurun_out/r05_pmc_mwalk.json \
    "prof_k2.py 1 GiB, 1024 patterns (4..12 B), k=1, count-only: the record walk" | cut -c1-400
rm -rf gpurun_out/r05_pmc_mwalk gpurun_out/r05_pmc_mwalk.*.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05_rec -o rec -- python $R/scripts/prof_records_r5.py 6 > $R/gpurun_out/r05_prof_records.log 2>&1)
f=$(find gpurun_out/prof_r05_rec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r05_records_kernel_stats.csv
rm -rf gpurun_out/prof_r05_rec
timeout 120 python scripts/perf_c5_worded_r5.py 2>&1 | grep "^c5" > gpurun_out/r05_perf_c5_worded_final.log
timeout 300 python scripts/file_stream_r4.py 4 2>&1 | grep -v "^/opt" > gpurun_out/r05_file_stream.log
bash scripts/startup_r5.sh > gpurun_out/r05_startup.log 2>&1
cut -c1-300 gpurun_out/r05_bench_line.json
tail -3 gpurun_out/r05_bench.err
grep "^c" gpurun_out/r05_prof_records.log
cat gpurun_out/r05_perf_c5_worded_final.log
tail -8 gpurun_out/r05_file_stream.log
grep -v "^\[agh" gpurun_out/r05_startup.log | tail -10
