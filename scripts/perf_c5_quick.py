"""Config 5 (1024 x 8..12 B, k = 1, count-only) on 4 GiB resident, the one-pass kernel only: device_ms median of 5.
usage: [AGH_LIB_PATH=...] scripts/perf_c5_quick.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import bench as B
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
pats, variants = B.c5_patterns_and_variants()
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=55, variants=variants, plant_period=500)
q = A.Query.multi(pats, k=1)
xs = []
for _ in range(6):
    r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP | A.TIME_SCAN)
    xs.append(r.device_ms)
xs = sorted(xs[1:])
print("c5 k=1 4 GiB (%s): device %.3f ms (%.0f GB/s) matched %d cand %d one-pass %d" %
      (os.path.basename(os.environ.get("AGH_LIB_PATH", "libagrep_hip.so")), xs[2], n / 1e6 / xs[2], r.n_matched, r.n_candidates, r.fused_segments))
