"""Where the one-pass -f kernel spends its time: AGH_MSCAN_DBG switches off level 3 (1), its text
loads (2), level 2 (4).  usage: scripts/perf_c5_breakdown_r4.py [GiB]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gib * (1 << 30))
t = torch.empty(n, dtype=torch.uint8, device='cuda')
rng = random.Random(1024)
ps = set()
while len(ps) < 1024:
    ps.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(8, 12))))
pats = sorted(ps)
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=5, variants=tuple(pats[:7]), plant_period=500)
for rb in ("13", "12"):
    for dbg, what in (("0", "everything"), ("2", "level 3 without text loads"), ("1", "no level 3"), ("5", "level 1 only")):
        os.environ["AGH_MSCAN_RB"] = rb
        os.environ["AGH_MSCAN_DBG"] = dbg
        q = A.Query.multi(pats, k=1)
        xs = []
        for _ in range(7):
            r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP | A.TIME_SCAN)
            xs.append(r.sweep_ms)
        xs.sort()
        print("rb=%s dbg=%s %-28s kernel %.3f ms (%.0f GB/s) matched %d cand %d" % (rb, dbg, what, xs[3], n / 1e6 / xs[3], r.n_matched, r.n_candidates), flush=True)
        q.close()
