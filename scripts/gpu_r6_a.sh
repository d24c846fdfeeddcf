#!/bin/bash
# Round 6, first GPU call: the tests of the hygiene changes, then the whole suite, then the bench line with the new
# legs (all_shards_nocase, c3.pinned_m29, engines).
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_records.py tests/test_gpu_refshim.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_a_pytest_touched.log
cat gpurun_out/r06_a_pytest_touched.log
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r06_a_bench.err | tail -1 > gpurun_out/r06_a_bench_line.json
tail -5 gpurun_out/r06_a_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_a_bench_line.json'))
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'])
cb=d['cpu_baseline']
for k in ('all_shards_records_sha256_equal','all_shards_nocase_records_sha256_equal','all_shards_nocase_error'):
    print(k,cb.get(k))
print('nocase',{k:v for k,v in cb.get('all_shards_nocase',{}).items() if k!='what'})
print('m29',{k:v for k,v in d.get('c3',{}).get('pinned_m29',{}).items() if k not in('workload',)})
for k,v in d.get('engines',{}).items():
    if k!='workload': print(k, {a:b for a,b in v.items() if a in('value','ms_per_step','engine','slice_equals_oracle','slice_count','error')}, v.get('roofline',{}).get('frac'))
print(d.get('engines_error'), d.get('configs_error'))
for c in ('c2_records','c3','c5','c5_as_worded'):
    print(c, d[c]['value'], d[c]['ms_per_step'], d[c]['roofline']['frac'])
PY
