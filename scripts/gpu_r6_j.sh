# (only delimiters that occur in the corpus: a text without its delimiter is ONE record, and the table engine walks a record serially)
# A/B: the delimiter-end bitmap by 64-byte words per lane (round 5, AGH_DBM_WORDS=1) against 1 KiB strips per wave
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in 1 0; do for d in 'e ' 'e  ' 'i   ' '   '; do
  echo "AGH_DBM_WORDS=$w delim '$d'"
  AGH_DBM_WORDS=$w timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/scripts/prof_table_delim_r6.py "$d" 1 > /tmp/o.log 2>&1
  grep -v "^[WIE]2026" /tmp/o.log | tail -1
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/p/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if 'k_delim_bitmap' in r['Name']:
        print('  %-60s calls %5s avg_us %10.1f'%(r['Name'][:60],r['Calls'],float(r['AverageNs'])/1e3))
PY
  rm -rf /tmp/p
done; done
