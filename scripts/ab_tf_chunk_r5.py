"""Table engine, fast form: bytes per lane (AGH_TF_CHUNK) against the length of the records, 4 GiB count-only."""
import os, sys
os.environ.setdefault("AGH_ENV_LIVE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import bench as B
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
for pat, k, delim in ((b"approx#match", 1, b"\n"), (b"approx#match", 1, b"e "), (b"approx#match", 1, b"s\n"), (b"approx#match", 1, b"hs\n")):
    row = []
    for c in ("0", "4096", "8192", "16384", "32768"):
        os.environ["AGH_TF_CHUNK"] = c
        with A.Query.pattern(pat, k, delim=delim) as q:
            xs = sorted(q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SCAN).device_ms for _ in range(5))
            r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.FORCE_NUMBERED)
        row.append("%s: %.3f" % (c, xs[2]))
    print("'%s' k=%d -d %r (%d records, %.0f B each)  count-only ms by chunk  %s" % (pat.decode(), k, delim, r.n_records, n / max(1, r.n_records), "  ".join(row)), flush=True)
