"""A/B inside one process: gram-offset-tight verify windows + false-positive rejection
(AGH_TIGHT_VERIFY=1, default) vs the offset-blind window.  usage: ab_verify.py [gib] [k] [m]"""
import os, sys, time
os.environ.setdefault("AGH_ENV_LIVE", "1")   # switches are flipped between scans of one query
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(gib * (1 << 30)) & ~4095
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
if len(sys.argv) > 3:       # config C3 shape: m = 48, -i
    import random
    rng = random.Random(48)
    pat = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(48))
    A.corpus_fill_device(t.data_ptr(), n // 4096, seed=9, variants=(pat, pat[:20] + b"Q" + pat[21:]), plant_period=500, upper_permille=500)
    q = A.Query(pat, k, nocase=True)
else:
    q = A.Query(b"approximatematch", k)
print(q.info())
for rnd in range(3):
    for mode in ("0", "1"):
        os.environ["AGH_TIGHT_VERIFY"] = mode
        for _ in range(3):
            r = q.scan_device(t.data_ptr(), n, flags=A.COUNT)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            r = q.scan_device(t.data_ptr(), n, flags=A.COUNT)
        dt = (time.perf_counter() - t0) / 30
        print("tight=%s  %.4f ms/scan  %.0f GB/s  sweep %.4f ms  device %.4f ms  matched %d cand %d"
              % (mode, dt * 1e3, n / 1e9 / dt, r.sweep_ms, r.device_ms, r.n_matched, r.n_candidates), flush=True)
