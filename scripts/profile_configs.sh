#!/bin/bash
# rocprofv3 kernel statistics of every configuration DESIGN.md quotes (run on the GPU box):
#   scripts/profile_configs.sh <tag>      -> gpurun_out/<tag>_*  (copy what is judged into profiles/)
# One rocprofv3 run per workload, --kernel-trace --stats only (no counters here; the PMC passes are
# scripts/pmc_passes.sh, separate runs as MI355X_MICROARCH.md prescribes).
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {   # name, command...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$name -o $name -- "$@" > $OUT/${TAG}_$name.log 2>&1
    local f=$(find $OUT/prof_${TAG}_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $OUT/${TAG}_${name}_kernel_stats.csv
    tail -3 $OUT/${TAG}_$name.log
}
prof bench     python $R/bench.py --steps 10 --warmup 2 --no-traffic --no-cpu-baseline
prof c3        python $R/scripts/config_runs.py c3
prof c5        python $R/scripts/config_runs.py c5 c5k
prof fallback  python $R/scripts/perf_fullscan.py 4
prof pieces    python $R/scripts/perf_short.py
ls $OUT | grep "${TAG}_.*kernel_stats" 
