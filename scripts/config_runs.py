"""BASELINE.json configs 3 and 5 at their stated single-GPU sizes (resident in HBM):
   c3: m=48 pattern, k=3, -i, 16 GiB (two 8 GiB segments);  c5: 1024 patterns, exact, 8 GiB
   (the per-GPU share of 32 GiB over 4 GPUs)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O

which = sys.argv[1:] or ["c3", "c5"]
if "c3" in which:
    rng = random.Random(48)
    pat = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(48))
    vs = [pat]
    for edits in (1, 2, 3, 4):
        v = bytearray(pat)
        for _ in range(edits):
            op, pos = rng.randint(0, 2), rng.randrange(4, len(v) - 4)
            if op == 0: v[pos] = ord("Q")
            elif op == 1: del v[pos]
            else: v.insert(pos, ord("Z"))
        vs.append(bytes(v))
    n = 16 << 30
    t = torch.empty(n, dtype=torch.uint8, device='cuda')
    planted = A.corpus_fill_device(t.data_ptr(), n // 4096, seed=9, variants=tuple(vs), plant_period=500, upper_permille=500)
    q = A.Query(pat, 3, nocase=True)
    print("c3 filter", q.info())
    for fl, lab in ((A.COUNT, "lean"), (0, "numbered")):
        xs = []
        for i in range(4):
            t0 = time.perf_counter(); r = q.scan_device(t.data_ptr(), n, flags=fl); xs.append(time.perf_counter() - t0)
        w = sorted(xs)[1]
        print("c3 m=48 k=3 -i 16 GiB %-8s wall %.3f ms  %.0f GB/s  matched %d (planted 0..3 edits: %d) records %d cand %d"
              % (lab, w * 1e3, n / 1e9 / w, r.n_matched, sum(planted[:4]), r.n_records, r.n_candidates))
    r_full = q.scan_device(t.data_ptr(), 2 << 30, flags=A.FORCE_FULLSCAN)
    r_filt = q.scan_device(t.data_ptr(), 2 << 30)
    print("c3 engines agree on the first 2 GiB:", r_full.n_matched == r_filt.n_matched, r_full.n_matched,
          "fullscan %.0f GB/s" % ((2 << 30) / 1e6 / r_full.device_ms))
    q.close(); del t; torch.cuda.empty_cache()
if "c5" in which:
    rng = random.Random(1024)
    pats = set()
    while len(pats) < 1024:
        pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(4, 12))))
    pats = sorted(pats)
    n = 8 << 30
    t = torch.empty(n, dtype=torch.uint8, device='cuda')
    A.corpus_fill_device(t.data_ptr(), n // 4096, seed=5, variants=tuple(pats[:7]), plant_period=500)
    q = A.Query.multi(pats)
    xs = []
    for i in range(4):
        t0 = time.perf_counter(); r = q.scan_device(t.data_ptr(), n, flags=A.FILENAMEONLY); xs.append(time.perf_counter() - t0)
    w = sorted(xs)[1]
    print("c5 1024 patterns exact 8 GiB -l wall %.3f ms  %.0f GB/s  matched records %d cand %d" % (w * 1e3, n / 1e9 / w, r.n_matched, r.n_candidates))
    q.close()
if "c5k" in which:
    # -f with k = 1 (the config as BASELINE words it).  Patterns of 4..5 bytes with one error
    # match almost every record; 8..12-byte patterns are the selective case.
    rng = random.Random(1024)
    for lo, hi, mib in ((8, 12, 1024), (4, 12, 256)):
        pats = set()
        while len(pats) < 1024:
            pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(lo, hi))))
        pats = sorted(pats)
        n = mib << 20
        t = torch.empty(n, dtype=torch.uint8, device='cuda')
        A.corpus_fill_device(t.data_ptr(), n // 4096, seed=5, variants=tuple(pats[:7]), plant_period=500)
        q = A.Query.multi(pats, k=1)
        xs = []
        for i in range(3):
            t0 = time.perf_counter(); r = q.scan_device(t.data_ptr(), n, flags=A.FILENAMEONLY); xs.append(time.perf_counter() - t0)
        w = sorted(xs)[1]
        print("c5k 1024 patterns len %d..%d k=1 %d MiB wall %.3f ms  %.1f GB/s  matched records %d cand %d"
              % (lo, hi, mib, w * 1e3, n / 1e9 / w, r.n_matched, r.n_candidates))
        q.close(); del t
