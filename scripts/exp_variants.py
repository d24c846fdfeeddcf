"""(needs the diagnostics build: make -C agrep_amd/csrc EXP=1.)  Attribute the gap between the read probe and the sweep: structural variants, interleaved."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
from agrep_amd import _ffi
import _oracle as O
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
names = {-1: "read probe (8 waves/SIMD, no prefetch)", 0: "bare loop, no prefetch", 16: "bare + prefetch",
         20: "+32K LDS alloc (5 waves/SIMD)", 17: "hash VALU, no LDS", 21: "hash VALU + LDS alloc",
         23: "hash + LDS lookups (32K)", 55: "hash + LDS lookups (16K table)", 7: "hash+lookups, no prefetch",
         24: "census VALU only", 31: "hash + lookups + census",
         64: "bare loop, nt loads", 80: "bare + prefetch, nt loads", 87: "hash + LDS lookups (32K), nt loads"}
res = {e: [] for e in names}
for rnd in range(6):
    for e in names:
        ms = _ffi.probe_variant_ms(t.data_ptr(), n, e)
        if rnd: res[e].append(ms)
for e, xs in res.items():
    xs.sort()
    print("%-45s min %.3f med %.3f ms  %.0f GB/s" % (names[e], xs[0], xs[len(xs)//2], n/1e6/xs[len(xs)//2]))
q = A.Query(O.PATTERN_C2, 2)
for fl, lab in ((A.COUNT, "real lean sweep"), (0, "real census sweep")):
    xs = sorted(q.scan_device(t.data_ptr(), n, flags=fl).sweep_ms for _ in range(6))
    print("%-45s min %.3f med %.3f ms" % (lab, xs[0], xs[3]))
