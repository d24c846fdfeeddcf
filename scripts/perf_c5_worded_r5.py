"""Config 5 as SURVEY 8d words it (1024 patterns of 4..12 bytes, k = 1, count-only) on 4 GiB resident: the record walk
(agh_mwalk.hip).  usage: scripts/perf_c5_worded_r5.py [GiB]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(gib * (1 << 30))
rng = random.Random(1024)
pats = set()
while len(pats) < 1024:
    pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(4, 12))))
pats = sorted(pats)
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=5, variants=tuple(pats[:7]), plant_period=500)
q = A.Query.multi(pats, k=1)
xs = []
for _ in range(5):
    r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP | A.TIME_SCAN)
    xs.append(r.device_ms)
xs = sorted(xs[1:])
rn = q.scan_device(t.data_ptr(), 256 << 20, flags=A.COUNT | A.FORCE_NUMBERED)
rl = q.scan_device(t.data_ptr(), 256 << 20, flags=A.COUNT)
print("c5 as worded (4..12 B, k=1) %.0f GiB: device %.3f ms (%.0f GB/s) matched %d one-pass %d | 256 MiB: count-only %d numbered %d"
      % (gib, xs[1], n / 1e6 / xs[1], r.n_matched, r.fused_segments, rl.n_matched, rn.n_matched))
