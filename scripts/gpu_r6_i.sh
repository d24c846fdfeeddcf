# A/B: bytes per lane of the fast table kernels (AGH_TF_CHUNK) by text size, 1.7 KB records and lines
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in 1 2 4 8; do for ch in 4096 8192 16384; do for d in 's\n' '\n'; do
  echo -n "GiB=$g AGH_TF_CHUNK=$ch "; AGH_TF_CHUNK=$ch python $R/scripts/prof_table_delim_r6.py "$d" 1 $g 2>/dev/null | tail -1
done; done; done
