"""Small driver for rocprofv3: N scans of the C2 workload (m=16, k from argv) resident in HBM."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mode = sys.argv[3] if len(sys.argv) > 3 else "lean"
flags = {"full": A.FORCE_FULLSCAN, "fullc": A.FORCE_FULLSCAN | A.COUNT, "numbered": 0, "lean": A.COUNT, "multi": A.COUNT,
         "multik": A.COUNT, "table": A.COUNT, "word": A.FORCE_FULLSCAN | A.COUNT}[mode]
n = int(gib * (1 << 30))
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
if mode == "multi":                      # the config-5 pattern set (1024 patterns of 4..12 bytes)
    import random
    rng = random.Random(1024)
    pats = set()
    while len(pats) < 1024:
        pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(4, 12))))
    q = A.Query.multi(sorted(pats), k=k)
elif mode == "multik":                   # config 5 with errors: 1024 patterns of 8..12 bytes, k from argv
    import random
    rng = random.Random(1024)
    pats = set()
    while len(pats) < 1024:
        pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(8, 12))))
    q = A.Query.multi(sorted(pats), k=k)
elif mode == "word":                     # `matching` (m = 8) with k errors on the fast full scan (three streams per lane)
    q = A.Query(b"matching", k)
elif mode == "table":                    # 'approx#match' on the reference's own tables (table engine)
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "pattern_language.json")))["cases"]
    c = [x for x in gold if x["pattern"] == "approx#match" and x["k"] == min(k, 1)][0]
    tb = c["tables"]
    q = A.Query.from_maskgen(tb["Mask"], tb["Init0"], tb["Init1"], tb["NO_ERR_MASK"], tb["endposition"],
                             tb["D_endpos"], tb["D_endpos"].bit_length(), b"\n", c["k"], tb["AND"])
else:
    q = A.Query(O.PATTERN_C2, k)
for it in range(6):
    r = q.scan_device(t.data_ptr(), n, flags=flags)
print("k", k, "matched", r.n_matched, "cand", r.n_candidates, "dev_ms %.3f sweep_ms %.3f" % (r.device_ms, r.sweep_ms))
