"""Short patterns / many errors (no sample filter): piece engine vs the full scan."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
for pat, k in ((b"approxim", 1), (b"approxim", 2), (b"match", 1), (b"matematch", 2), (b"appr", 0), (b"approximate", 3),
               (b"approximatematch", 3), (b"approximatematch", 4)):
    q = A.Query(pat, k)
    row = []
    for fl, lab in ((A.COUNT, "default"), (A.COUNT | A.FORCE_FULLSCAN, "fullscan")):
        for _ in range(2):
            r = q.scan_device(t.data_ptr(), n, flags=fl)
        t0 = time.perf_counter()
        for _ in range(5):
            r = q.scan_device(t.data_ptr(), n, flags=fl)
        dt = (time.perf_counter() - t0) / 5
        row.append("%s %.3f ms %.0f GB/s matched %d cand %d engine %d" % (lab, dt * 1e3, n / 1e9 / dt, r.n_matched, r.n_candidates, r.engine))
    print(pat.decode(), "k=%d" % k, q.info(), " | ".join(row), flush=True)
    q.close()
