"""BASELINE config 5 on resident text: 1024 patterns, -f, count-only -- the one-pass kernel (agh_mscan.hip)
against the two-kernel form (AGH_MSCAN=0), both table sizes.  device_ms = HIP events around the scan's
kernel sequence, sweep_ms = the kernel that reads every byte.
usage: scripts/perf_c5_r4.py [GiB, default 4] [reps, default 7]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
n_all = int(gib * (1 << 30))
t = torch.empty(n_all, dtype=torch.uint8, device='cuda')


def pats_of(npat, lo, hi, seed=1024):
    rng = random.Random(seed)
    ps = set()
    while len(ps) < npat:
        ps.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(lo, hi))))
    return sorted(ps)


def run(label, pats, k, n, flags, env):
    for key in ("AGH_MSCAN", "AGH_MSCAN_RB"):
        os.environ.pop(key, None)
    os.environ.update(env)
    A.corpus_fill_device(t.data_ptr(), n // 4096, seed=5, variants=tuple(pats[:7]), plant_period=500)
    q = A.Query.multi(pats, k=k)
    xs = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = q.scan_device(t.data_ptr(), n, flags=flags)
        xs.append((time.perf_counter() - t0, r.device_ms, r.sweep_ms))
    xs.sort()
    w, d, s = xs[len(xs) // 2]
    print("%-34s %-22s k=%d %6.0f MiB: wall %.3f ms (%.0f GB/s)  device %.3f ms (%.0f GB/s)  sweep %.3f ms (%.0f GB/s)  "
          "matched %d cand %d segs %d one-pass %d reruns %d" % (label, " ".join("%s=%s" % kv for kv in env.items()) or "default", k,
                                                              n / 2**20, w * 1e3, n / 1e9 / w, d, n / 1e6 / max(d, 1e-9), s,
                                                              n / 1e6 / max(s, 1e-9), r.n_matched, r.n_candidates, r.n_segments,
                                                              r.fused_segments, r.lean_reruns), flush=True)
    q.close()
    return r.n_matched


F = A.COUNT | A.TIME_SWEEP | A.TIME_SCAN
for label, lo, hi, k in (("1024 x 8..12 B k=1 (config 5)", 8, 12, 1), ("1024 exact 8..12 B", 8, 12, 0),
                         ("1024 exact 4..12 B", 4, 12, 0), ("1024 exact 5..12 B", 5, 12, 0),
                         ("1024 x 8..14 B k=1", 8, 14, 1), ("64 x 8..12 B k=1", 8, 12, 1)):
    pats = pats_of(64 if label.startswith("64") else 1024, lo, hi)
    got = [run(label, pats, k, n_all, F, env) for env in ({"AGH_MSCAN": "0"}, {"AGH_MSCAN_RB": "13"}, {"AGH_MSCAN_RB": "12"})]
    assert got[0] == got[1] == got[2], got
