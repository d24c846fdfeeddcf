"""A/B the k_sweep launch variants (AGH_SWEEP_VARIANT) in one process, interleaved rounds.
The env hook is not in the library by default: it is a ten-line dispatch over
k_sweep<H, MODE, BLOCK, PREFETCH> in launch_sweep_hm (agh_sweep.hip) that was added for the
tuning rounds recorded in DESIGN.md (d) and removed again."""
import os
os.environ.setdefault("AGH_ENV_LIVE", "1")   # switches are flipped between scans of one query
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O

n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
variants = sys.argv[1:] or ["256", "256p", "512", "512p", "1024", "1024p"]
for k in (2, 0):
    q = A.Query(O.PATTERN_C2, k)
    stats = {v: [] for v in variants}
    tot = {v: [] for v in variants}
    for rnd in range(7):
        for v in variants:
            os.environ["AGH_SWEEP_VARIANT"] = v
            r = q.scan_device(t.data_ptr(), n)
            if rnd:
                stats[v].append(r.sweep_ms); tot[v].append(r.device_ms)
    for v in variants:
        a = sorted(stats[v]); b = sorted(tot[v])
        print("k=%d variant %-6s sweep ms min %.3f med %.3f  (%.0f GB/s)   total ms med %.3f (%.0f GB/s) matched %d"
              % (k, v, a[0], a[len(a)//2], n/1e6/a[len(a)//2], b[len(b)//2], n/1e6/b[len(b)//2], r.n_matched))
    q.close()
print("probe", [round(n/1e6/A.probe_read_ms(t.data_ptr(), n)) for _ in range(3)])
