"""One table query under a two-byte delimiter on 4 GiB, for rocprofv3 --kernel-trace --stats."""
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
with A.Query.pattern(b"approx#match", 1, delim=sys.argv[1].encode() if len(sys.argv) > 1 else b"e ") as q:
    for _ in range(5):
        r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SCAN)
    print(r.device_ms, r.n_matched)
