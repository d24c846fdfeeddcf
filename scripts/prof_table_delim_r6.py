"""rocprofv3 driver: 'approx#match' on the table engine, 4 GiB resident, count-only, with the delimiter from argv
(default 'e ': a delimiter of two bytes -> the delimiter-end bitmap in front of the scan)."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O

delim = (sys.argv[1] if len(sys.argv) > 1 else "e ").encode().decode("unicode_escape").encode("latin1")
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(float(sys.argv[3]) * (1 << 30)) if len(sys.argv) > 3 else 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
q = A.Query.pattern(b"approx#match", k, delim=delim)
for it in range(6):
    r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SCAN | A.TIME_SWEEP)
print("delim", delim, "k", k, "matched", r.n_matched, "dev_ms %.3f sweep_ms %.3f" % (r.device_ms, r.sweep_ms))
