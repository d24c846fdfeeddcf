"""The sample-filter pipelines under delimiters of several bytes / a folded letter on 4 GiB (device_ms includes the
delimiter-end bitmap since round 5).  usage: scripts/perf_mb_filter_r5.py [GiB]"""
import os, sys
os.environ.setdefault("AGH_ENV_LIVE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import bench as B
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gib * (1 << 30))
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)


def med(q, flags, reps=5):
    xs = []
    for _ in range(reps):
        r = q.scan_device(t.data_ptr(), n, flags=flags)
        xs.append(r.device_ms if r.device_ms > 0 else r.sweep_ms)
    return sorted(xs)[reps // 2], r


for delim, nocase in ((b"\n", False), (b"e ", False), (b"s\n", False), (b"\n\n", False), (b"z", True)):
    with A.Query(B.PATTERN, 2, nocase=nocase, delim=delim) as q:
        ms_c, r_c = med(q, A.COUNT | A.TIME_SCAN | A.TIME_SWEEP)
        ms_n, r_n = med(q, A.COUNT | A.FORCE_NUMBERED | A.TIME_SCAN, 3)
        ms_l, r_l = med(q, A.TIME_SCAN, 3)
    print("'%s' k=2 -d %r%s  count-only %.3f ms %.0f GB/s | numbered count %.3f ms | match list %.3f ms (matched %d/%d/%d, records %d)"
          % (B.PATTERN.decode(), delim, " -i" if nocase else "", ms_c, n / 1e6 / max(ms_c, 1e-6), ms_n, ms_l, r_c.n_matched, r_n.n_matched, r_l.n_matched, r_n.n_records), flush=True)
