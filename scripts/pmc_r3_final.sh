#!/bin/bash
# PMC summaries of the kernels that changed after scripts/pmc_r3.sh ran: the fast table engine and
# -f with one error (side check in two 64-bit words).  Same passes, same summary script.
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
scripts/pmc_passes.sh pmc_r03_multik 2 1 multik > gpurun_out/r03_pmc_multik.log 2>&1
scripts/pmc_passes.sh pmc_r03_table 2 0 table > gpurun_out/r03_pmc_table.log 2>&1
python scripts/pmc_summary.py pmc_r03_multik k_sweep_multi 2147483648 gpurun_out/r03_pmc_sweep_multi_k1.json "prof_k2.py 2 GiB, 1024 patterns (8..12 B), k=1, count-only"
python scripts/pmc_summary.py pmc_r03_multik k_verify_multi 2147483648 gpurun_out/r03_pmc_verify_multi_k1.json "prof_k2.py 2 GiB, 1024 patterns (8..12 B), k=1, count-only"
python scripts/pmc_summary.py pmc_r03_table k_tablescan_fast 2147483648 gpurun_out/r03_pmc_tablescan_fast.json "prof_k2.py 2 GiB, 'approx#match' k=0 on the reference's tables, count-only, fast form"
ls gpurun_out | grep "^r03_pmc"
# the headline kernel itself (8 GiB segment = what a rank holds at N = 8; k = 2 and k = 0)
scripts/pmc_passes.sh pmc_r03_lean 8 2 lean > gpurun_out/r03_pmc_lean.log 2>&1
scripts/pmc_passes.sh pmc_r03_lean_k0 8 0 lean > gpurun_out/r03_pmc_lean_k0.log 2>&1
python scripts/pmc_summary.py pmc_r03_lean k_sweep_fused 8589934592 gpurun_out/r03_pmc_sweep_fused.json "prof_k2.py 8 GiB, m=16 k=2, count-only: the fused kernel (H = 2 samples, three workgroups per CU)"
python scripts/pmc_summary.py pmc_r03_lean_k0 k_sweep_fused 8589934592 gpurun_out/r03_pmc_sweep_fused_k0.json "prof_k2.py 8 GiB, m=16 k=0, count-only: the fused kernel (H = 8 samples, two workgroups per CU)"
