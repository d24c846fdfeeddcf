#!/bin/bash
# Round-3 PMC summaries (one GPU call): the kernels VERDICT r02 asked about --
#   fast full scan (k = 2), multi-pattern sweep + verifier (exact 4..12 B and k = 1 over 8..12 B), table engine.
# Four separate rocprofv3 --pmc passes per workload (scripts/pmc_passes.sh), --kernel-trace only.
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
(rocprofv3 -L 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) | grep -iE "TCC_EA0_RDREQ|FETCH_SIZE|TCC_HIT|TCC_MISS|TCC_REQ" | head -40 > gpurun_out/r03_counter_names.log
scripts/pmc_passes.sh pmc_r03_fullc 2 2 fullc > gpurun_out/r03_pmc_fullc.log 2>&1
scripts/pmc_passes.sh pmc_r03_multi 2 0 multi > gpurun_out/r03_pmc_multi.log 2>&1
scripts/pmc_passes.sh pmc_r03_multik 2 1 multik > gpurun_out/r03_pmc_multik.log 2>&1
# (r03_pmc_sweep_multi*_fused.json: the same two workloads with AGH_MP_FUSED=1 at commit 97f45c9, where a full
#  queue was verified inside the sweeping wave -- measured slower and removed since)
scripts/pmc_passes.sh pmc_r03_table 2 0 table > gpurun_out/r03_pmc_table.log 2>&1
python scripts/pmc_summary.py pmc_r03_fullc k_fullscan_fast 2147483648 gpurun_out/r03_pmc_fullscan_fast.json "prof_k2.py 2 GiB, m=16 k=2, count-only full scan, fast form (two streams per lane)"
python scripts/pmc_summary.py pmc_r03_multi k_sweep_multi 2147483648 gpurun_out/r03_pmc_sweep_multi.json "prof_k2.py 2 GiB, 1024 exact patterns (4..12 B), count-only"
python scripts/pmc_summary.py pmc_r03_multi k_verify_multi 2147483648 gpurun_out/r03_pmc_verify_multi_exact.json "prof_k2.py 2 GiB, 1024 exact patterns (4..12 B), count-only"
python scripts/pmc_summary.py pmc_r03_multik k_sweep_multi 2147483648 gpurun_out/r03_pmc_sweep_multi_k1.json "prof_k2.py 2 GiB, 1024 patterns (8..12 B), k=1, count-only"
python scripts/pmc_summary.py pmc_r03_multik k_verify_multi 2147483648 gpurun_out/r03_pmc_verify_multi_k1.json "prof_k2.py 2 GiB, 1024 patterns (8..12 B), k=1, count-only"
python scripts/pmc_summary.py pmc_r03_table k_tablescan_fast 2147483648 gpurun_out/r03_pmc_tablescan_fast.json "prof_k2.py 2 GiB, 'approx#match' k=0 on the reference's tables, count-only, fast form"
ls gpurun_out | grep "^r03_pmc"
