# A/B of the table engine's hand-over threshold on the long-record case (delimiter 's\n': records of ~1.7 KB)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in ${CONTS:-0 8 16 32 48}; do
  echo "AGH_TF_CONT=$c"
  AGH_TF_CONT=$c rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/scripts/prof_table_delim_r6.py 's\n' 1 > /tmp/o.log 2>&1
  grep -v "^[WIE]2026" /tmp/o.log | tail -1
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/p/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    if any(x in r['Name'] for x in ('k_corpus','copyBuffer','fillBuffer','hashset')): continue
    print('  %-70s calls %5s avg_us %10.1f pct %5s'%(r['Name'][:70],r['Calls'],float(r['AverageNs'])/1e3,r['Percentage']))
PY
  rm -rf /tmp/p
done
