#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/perf_c5_worded_r6.py 4 2>&1 | grep "^c5" > gpurun_out/r06_ab_mtile_phases.log
cat gpurun_out/r06_ab_mtile_phases.log
bash scripts/pmc_passes.sh r06_pmc_mtile 1 1 multi > /dev/null 2>&1
python scripts/pmc_summary.py r06_pmc_mtile "k_mtile<" 1073741824 gpurun_out/r06_pmc_mtile_v1.json \
    "prof_k2.py 1 GiB, 1024 patterns (4..12 B), k=1, count-only: the tile kernel, first version" | cut -c1-1500
rm -rf gpurun_out/r06_pmc_mtile gpurun_out/r06_pmc_mtile.*.log
