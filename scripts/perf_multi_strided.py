"""Exact multi-pattern scan by length class: the probe stride the host picks (1, 2 or 4) and what
it buys.  4 GiB resident corpus, 1024 patterns, count-only."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
rng = random.Random(1024)
for lo, hi in ((8, 12), (5, 12), (7, 12), (4, 12)):
    pats = set()
    while len(pats) < 1024:
        pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(lo, hi))))
    pats = sorted(pats)
    n = 4 << 30
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    A.corpus_fill_device(t.data_ptr(), n // 4096, seed=5, variants=tuple(pats[:7]), plant_period=500)
    q = A.Query.multi(pats)
    xs = []
    for i in range(5):
        r = q.scan_device(t.data_ptr(), n, flags=A.COUNT)
        xs.append((r.device_ms, r.sweep_ms))
    xs.sort()
    d, s = xs[2]
    print("1024 exact patterns %d..%d B, stride %d: device %.3f ms (%.0f GB/s) sweep %.3f ms (%.0f GB/s) matched %d cand %d"
          % (lo, hi, q.info()["filter_h"], d, n / 1e6 / d, s, n / 1e6 / s, r.n_matched, r.n_candidates), flush=True)
    q.close()
    del t
