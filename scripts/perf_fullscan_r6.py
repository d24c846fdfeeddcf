"""The fast full scan by text streams per lane (AGH_FS_STREAMS): `matching` (m = 8) and a 10-byte word, k = 1..3, 4 GiB of the
bench corpus resident, count-only.  usage: scripts/perf_fullscan_r6.py"""
import os, sys
os.environ["AGH_ENV_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
for pat in (b"matching", b"wordlength"):
    for k in (1, 2, 3):
        for streams in ("2", "0"):
            os.environ["AGH_FS_STREAMS"] = streams
            with A.Query(pat, k) as q:
                xs = []
                for _ in range(4):
                    r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.FORCE_FULLSCAN | A.TIME_SCAN)
                    xs.append(r.device_ms)
            xs = sorted(xs[1:])
            print("fullscan %s k=%d AGH_FS_STREAMS=%s: device %.3f ms (%.0f GB/s) matched %d" % (pat.decode(), k, streams, xs[1], n / 1e6 / xs[1], r.n_matched), flush=True)
