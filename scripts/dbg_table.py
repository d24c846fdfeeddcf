import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import agrep_amd as agh
import _oracle as O
rng = np.random.default_rng(7)
alpha = np.frombuffer(b"acrest\n", dtype=np.uint8)
t = alpha[rng.integers(0, len(alpha), 200000)].tobytes()
cases = json.load(open(os.path.join(ROOT, "tests/golden/pattern_language.json")))["cases"]
case = [c for c in cases if c["pattern"] == "car;red"][0]
tb = case["tables"]; M = tb["D_endpos"].bit_length()
ot = O.tables_from_golden(tb, M)
want = O.asearch_tables(ot, 1, t)[0]
for trial in range(3):
    q = agh.Query.from_maskgen(tb["Mask"], tb["Init0"], tb["Init1"], tb["NO_ERR_MASK"], tb["endposition"], tb["D_endpos"], M, b"\n", 1, tb["AND"])
    out = []
    seq = [(200000, 0), (0, agh.COUNT), (0, agh.COUNT), (200000, 0), (0, agh.COUNT), (0, 0), (0, 0), (0, agh.COUNT)] if trial < 2 else [(0, agh.COUNT)] * 8
    for cap, fl in seq:
        r, _ = q.scan_buffer(t, cap=cap, flags=fl)
        out.append((r.n_matched, r.n_records))
    print(trial, want, out)
    q.close()
