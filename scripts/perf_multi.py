"""Config-5 shape: 1024 patterns (4..12 bytes), exact, -l/-c style count on a resident corpus."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(gib * (1 << 30))
rng = random.Random(1024)
pats = set()
while len(pats) < 1024:
    pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(4, 12))))
pats = sorted(pats)
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=5, variants=tuple(pats[:7]), plant_period=500)
for npat in (1024, 64, 8):
    q = A.Query.multi(pats[:npat])
    for fl, lab in ((A.COUNT, "lean"), (0, "numbered")):
        xs = []
        for i in range(5):
            r = q.scan_device(t.data_ptr(), n, flags=fl)
            xs.append((r.device_ms, r.sweep_ms))
        xs.sort()
        d, s = xs[2]
        print("%4d patterns %-8s device %.3f ms (%.0f GB/s)  sweep %.3f ms (%.0f GB/s)  matched %d cand %d"
              % (npat, lab, d, n / 1e6 / d, s, n / 1e6 / s, r.n_matched, r.n_candidates))
    q.close()
