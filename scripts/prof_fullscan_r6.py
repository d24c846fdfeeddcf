"""rocprofv3 driver: `matching` k from argv on 4 GiB, forced full scan, count-only, AGH_FS_STREAMS from the environment"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
with A.Query(b"matching", k) as q:
    for _ in range(5):
        r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.FORCE_FULLSCAN | A.TIME_SCAN)
print("matching k=%d streams=%s device %.3f ms matched %d" % (k, os.environ.get("AGH_FS_STREAMS", "0"), r.device_ms, r.n_matched))
