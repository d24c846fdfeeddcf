"""(needs the diagnostics build: make -C agrep_amd/csrc EXP=1.)  Does a verifier trailing the sweep find
its lines in L2 / MALL?  The sweep-shaped read loop plus two lanes per wave re-reading a chained pair of
128-byte lines `lag` supertiles (4 KiB steps of the wave) behind the stream position."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
from agrep_amd import _ffi
import _oracle as O
n = int(os.environ.get("AGH_EXP_GIB", "16")) << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
cases = []
for nm, base in (("nt", 64 + 16 + 1 + 2 + 4), ("plain", 16 + 1 + 2 + 4)):
    cases += [(nm + " no re-read", base)] + [(nm + (" lag %3d" % l if l != 255 else " far"), base + 128 + (l << 8))
                                              for l in (0, 1, 2, 4, 8, 16, 32, 255)]
res = {c: [] for c, _ in cases}
for rnd in range(7):
    for c, e in cases:
        ms = _ffi.probe_variant_ms(t.data_ptr(), n, e)
        if rnd: res[c].append(ms)
for c, xs in res.items():
    xs.sort()
    print("%-18s min %.3f med %.3f ms  %.0f GB/s" % (c, xs[0], xs[len(xs)//2], n/1e6/xs[len(xs)//2]), flush=True)
