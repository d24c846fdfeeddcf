"""Config 5 as SURVEY 8d words it (1024 patterns of 4..12 bytes, k = 1, count-only) on resident text: the tile kernel
(agh_mtile.hip) with 1 / 2 / 4 tiles per wave, same process, interleaved (round 5's record walk, removed after the first
A/B: 20.8 ms per 4 GiB, profiles/r06_ab_mtile_v1.log).
usage: AGH_ENV_LIVE=1 scripts/perf_c5_worded_r6.py [GiB]"""
import os, random, sys
os.environ["AGH_ENV_LIVE"] = "1"
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
rn = q.scan_device(t.data_ptr(), 256 << 20, flags=A.COUNT | A.FORCE_NUMBERED)
for rnd in range(2):
    for mt in ("1", "2", "4"):
        os.environ["AGH_MTILE"] = mt
        xs = []
        for _ in range(4):
            r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP | A.TIME_SCAN)
            xs.append(r.device_ms)
        xs = sorted(xs[1:])
        rl = q.scan_device(t.data_ptr(), 256 << 20, flags=A.COUNT)
        print("c5 as worded (4..12 B, k=1) %.0f GiB AGH_MTILE=%s: device %.3f ms (%.0f GB/s) matched %d examined %d one-pass %d reruns %d | 256 MiB: count-only %d numbered %d %s"
              % (gib, mt, xs[1], n / 1e6 / xs[1], r.n_matched, r.n_candidates, r.fused_segments, r.lean_reruns, rl.n_matched, rn.n_matched,
                 "OK" if rl.n_matched == rn.n_matched else "MISMATCH"), flush=True)
# what the phases cost (measurement switches of the kernel, AGH_MTILE_DBG): 1 = no walk over the candidate bits
# (phase A + the count), 2 = the walk without its text loads
for mt in ("1", "2", "4"):
    for dbg in ("1", "2", "4"):
        os.environ["AGH_MTILE"] = mt
        os.environ["AGH_MTILE_DBG"] = dbg
        xs = []
        for _ in range(4):
            r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP | A.TIME_SCAN)
            xs.append(r.device_ms)
        xs = sorted(xs[1:])
        print("c5 as worded %.0f GiB AGH_MTILE=%s AGH_MTILE_DBG=%s: device %.3f ms (matched %d%s; counter %d%s)"
              % (gib, mt, dbg, xs[1], r.n_matched, "" if dbg == "4" else ": not a count", r.n_candidates, ": rounds of the walk, summed over the waves" if dbg == "4" else ""), flush=True)
os.environ.pop("AGH_MTILE_DBG")
# the share-out threshold: lanes with candidates of their own at which the rest goes into the shared list (0: never)
for mt in ("2", "4"):
    for share in ("0", "16", "24", "32", "40", "48", "64"):
        os.environ["AGH_MTILE"] = mt
        os.environ["AGH_MTILE_SHARE"] = share
        xs = []
        for _ in range(4):
            r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP | A.TIME_SCAN)
            xs.append(r.device_ms)
        xs = sorted(xs[1:])
        print("c5 as worded %.0f GiB AGH_MTILE=%s AGH_MTILE_SHARE=%s: device %.3f ms (%.0f GB/s) matched %d examined %d"
              % (gib, mt, share, xs[1], n / 1e6 / xs[1], r.n_matched, r.n_candidates), flush=True)
os.environ.pop("AGH_MTILE_SHARE")
# record-returning (numbered) scans of the same set: the tile kernel's numbered form against round 5's k_dense_multi
os.environ["AGH_MTILE"] = "4"
nn = 1 << 30
for numbered in ("1", "0"):
    os.environ["AGH_MTILE_NUMBERED"] = numbered
    xs = []
    for _ in range(3):
        r = q.scan_device(t.data_ptr(), nn, flags=A.COUNT | A.FORCE_NUMBERED | A.TIME_SCAN)
        xs.append(r.device_ms)
    print("c5 as worded 1 GiB numbered (record numbers, bitmap) AGH_MTILE_NUMBERED=%s: device %.3f ms (%.0f GB/s) matched %d"
          % (numbered, min(xs), nn / 1e6 / min(xs), r.n_matched), flush=True)
os.environ.pop("AGH_MTILE_NUMBERED")
