"""The fallback engines on the 4 GiB C2 corpus: k_fullscan (automaton over every byte, the
asearch.c shape) for k = 0..3 and a 64-bit-word pattern, and k_tablescan (record-parallel, the
reference's own '#' / ';' tables from tests/golden/pattern_language.json).  device_ms = HIP events
around the whole kernel sequence (census sweep + prefix scan + engine + count)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import bench as B

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gib * (1 << 30))
t = torch.empty(n, dtype=torch.uint8, device='cuda')
planted = A.corpus_fill_device(t.data_ptr(), n // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)

def med(q, flags, reps=5):
    xs = []
    for _ in range(reps):
        r = q.scan_device(t.data_ptr(), n, flags=flags)
        xs.append(r.device_ms)
    return sorted(xs)[reps // 2], r

for k in (0, 1, 2, 3):
    with A.Query(B.PATTERN, k) as q:
        ms, r = med(q, A.FORCE_FULLSCAN)
        ms_f, r_f = med(q, 0)
        ms_l, r_l = med(q, A.FORCE_FULLSCAN | A.COUNT)
        print("fullscan m=16 k=%d: %.3f ms  %.0f GB/s  matched %d | count-only (no census): %.3f ms  %.0f GB/s  matched %d "
              "reruns %d | (filter engine: %.3f ms %.0f GB/s matched %d)"
              % (k, ms, n / 1e6 / ms, r.n_matched, ms_l, n / 1e6 / ms_l, r_l.n_matched, r_l.lean_reruns,
                 ms_f, n / 1e6 / ms_f, r_f.n_matched), flush=True)
with A.Query(b"approximatematchapproximatematchapproximatemat", 3, nocase=True) as q:
    ms, r = med(q, A.FORCE_FULLSCAN, 3)
    print("fullscan m=46 k=3 -i (64-bit words): %.3f ms  %.0f GB/s  matched %d" % (ms, n / 1e6 / ms, r.n_matched), flush=True)
with A.Query(B.PATTERN, 2).set_costs(2, 1, 1) as q:
    ms, r = med(q, A.FORCE_FULLSCAN, 3)
    print("fullscan m=16 k=2 costs I2 S1 D1 (general automaton): %.3f ms  %.0f GB/s  matched %d" % (ms, n / 1e6 / ms, r.n_matched), flush=True)
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "pattern_language.json")))["cases"]
for c in gold:
    if c["pattern"] in ("approx#match", "approxi;matematch", "scar,cat") and c["k"] <= 1:
        tb = c["tables"]
        M = tb["D_endpos"].bit_length()
        q = A.Query.from_maskgen(tb["Mask"], tb["Init0"], tb["Init1"], tb["NO_ERR_MASK"], tb["endposition"],
                                 tb["D_endpos"], M, b"\n", c["k"], tb["AND"])
        ms, r = med(q, 0, 3)
        print("tablescan '%s' k=%d: %.3f ms  %.0f GB/s  matched %d" % (c["pattern"], c["k"], ms, n / 1e6 / ms, r.n_matched), flush=True)
        q.close()
