"""One-pass -f kernel under AGH_MSCAN_DBG values given on the command line (measurement switches of
agh_mscan.hip).  usage: scripts/perf_c5_dbg_r4.py GiB dbg [dbg ...]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
n = int(float(sys.argv[1]) * (1 << 30))
t = torch.empty(n, dtype=torch.uint8, device='cuda')
rng = random.Random(1024)
ps = set()
while len(ps) < 1024:
    ps.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(8, 12))))
pats = sorted(ps)
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=5, variants=tuple(pats[:7]), plant_period=500)
for dbg in sys.argv[2:]:
    os.environ["AGH_MSCAN_DBG"] = dbg
    q = A.Query.multi(pats, k=1)
    xs = []
    for _ in range(7):
        r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP | A.TIME_SCAN)
        xs.append(r.sweep_ms)
    xs.sort()
    print("dbg=%s kernel %.3f ms (%.0f GB/s) matched %d cand %d" % (dbg, xs[3], n / 1e6 / xs[3], r.n_matched, r.n_candidates), flush=True)
    q.close()
