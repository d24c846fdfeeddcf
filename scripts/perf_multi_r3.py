"""-f engines on resident text: exact sets by probe stride, -f with errors (sparse and dense), dense
exact sets.  device_ms = HIP events around the whole kernel sequence of the scan.
usage: scripts/perf_multi_r3.py [GiB, default 4]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n_all = int(gib * (1 << 30))
t = torch.empty(n_all, dtype=torch.uint8, device='cuda')


def pats_of(npat, lo, hi, seed=1024):
    rng = random.Random(seed)
    ps = set()
    while len(ps) < npat:
        ps.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(lo, hi))))
    return sorted(ps)


def run(label, pats, k, n, flags, reps=5):
    A.corpus_fill_device(t.data_ptr(), n // 4096, seed=5, variants=tuple(pats[:7]), plant_period=500)
    q = A.Query.multi(pats, k=k)
    info = q.info()
    xs = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = q.scan_device(t.data_ptr(), n, flags=flags)
        xs.append((time.perf_counter() - t0, r.device_ms, r.sweep_ms))
    xs.sort()
    w, d, s = xs[len(xs) // 2]
    print("%-44s k=%d %6.0f MiB stride %d q %d: wall %.3f ms (%.0f GB/s)  device %.3f ms (%.0f GB/s)  sweep %.3f ms (%.0f GB/s)  "
          "matched %d cand %d segs %d reruns %d" % (label, k, n / 2**20, info["filter_h"], info["filter_q"], w * 1e3, n / 1e9 / w,
                                                    d, n / 1e6 / max(d, 1e-9), s, n / 1e6 / max(s, 1e-9), r.n_matched, r.n_candidates,
                                                    r.n_segments, r.lean_reruns), flush=True)
    q.close()


for lo, hi in ((4, 12), (5, 12), (7, 12), (8, 12)):
    for fl, lab in ((A.COUNT | A.TIME_SWEEP, "count"), (A.TIME_SWEEP, "numbered")):
        run("1024 exact %d..%d B %s" % (lo, hi, lab), pats_of(1024, lo, hi), 0, n_all, fl)
run("1024 x 8..12 B k=1 count", pats_of(1024, 8, 12), 1, n_all, A.COUNT | A.TIME_SWEEP)
run("1024 x 8..12 B k=1 -l", pats_of(1024, 8, 12), 1, n_all, A.FILENAMEONLY)
run("1024 x 8..12 B k=1 numbered", pats_of(1024, 8, 12), 1, min(n_all, 1 << 30), A.TIME_SWEEP)
run("1024 x 10..16 B k=1 count", pats_of(1024, 10, 16), 1, n_all, A.COUNT | A.TIME_SWEEP)
run("1024 x 12..20 B k=2 count", pats_of(1024, 12, 20), 2, n_all, A.COUNT | A.TIME_SWEEP)
run("64 x 8..12 B k=1 count", pats_of(64, 8, 12), 1, n_all, A.COUNT | A.TIME_SWEEP)
run("1024 x 4..12 B k=1 count (dense)", pats_of(1024, 4, 12), 1, 256 << 20, A.COUNT)
run("1024 x 4..12 B k=1 numbered (dense)", pats_of(1024, 4, 12), 1, 256 << 20, 0)
run("300 x 1..3 B exact count (dense)", pats_of(300, 1, 3), 0, 256 << 20, A.COUNT)
run("300 x 1..3 B exact numbered (dense)", pats_of(300, 1, 3), 0, 256 << 20, 0)
