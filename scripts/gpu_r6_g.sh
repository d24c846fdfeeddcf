cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 'e ' '\n' 's\n'; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/scripts/prof_table_delim_r6.py "$d" 1 > /tmp/o.log 2>&1
  grep -v "^[WIE]2026" /tmp/o.log | tail -8
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/p/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print('  %-70s calls %5s avg_us %10.1f pct %5s'%(r['Name'][:70],r['Calls'],float(r['AverageNs'])/1e3,r['Percentage']))
PY
  rm -rf /tmp/p
done
