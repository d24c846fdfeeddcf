"""Record walk (agh_mwalk.hip) on config 5 as worded: text bytes per lane (AGH_MW_CH), 4 GiB count-only."""
import os, random, sys
os.environ.setdefault("AGH_ENV_LIVE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
n = 4 << 30
rng = random.Random(1024)
pats = set()
while len(pats) < 1024:
    pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(4, 12))))
pats = sorted(pats)
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=5, variants=tuple(pats[:7]), plant_period=500)
for rnd in range(2):
    for ch in ("1024", "2048", "4096", "8192", "16384", "32768"):
        os.environ["AGH_MW_CH"] = ch
        with A.Query.multi(pats, k=1) as q:
            xs = sorted(q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP | A.TIME_SCAN).device_ms for _ in range(4))
            r = q.scan_device(t.data_ptr(), n, flags=A.COUNT)
        print("round %d  AGH_MW_CH=%s: device %.3f ms (%.0f GB/s) matched %d one-pass %d" % (rnd, ch, xs[1], n / 1e6 / xs[1], r.n_matched, r.fused_segments), flush=True)
