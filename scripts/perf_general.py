"""Edit costs / <exact> / -w queries: sample filter + general-automaton verify vs the full scan."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
cases = json.load(open(os.path.join(ROOT, "tests/golden/pattern_language.json")))["cases"]
qs = []
q = A.Query(O.PATTERN_C2, 2); q.set_costs(2, 1, 1); qs.append(("costs I2 S1 D1 k=2", q))
q = A.Query(O.PATTERN_C2, 3); q.set_costs(1, 2, 3); qs.append(("costs I1 S2 D3 k=3", q))
for c in cases:
    if c["pattern"] == "approximatematch" and c["opts"] in (["-w"], ["-x"]) or c["pattern"] == "appr[ox]ximatematch":
        tb = c["tables"]; M = tb["D_endpos"].bit_length()
        qs.append(("%s %s k=%d" % (c["pattern"], "".join(c["opts"]), c["k"]),
                   A.Query.from_maskgen(tb["Mask"], tb["Init0"], tb["Init1"], tb["NO_ERR_MASK"], tb["endposition"],
                                        tb["D_endpos"], M, b"\n", c["k"], tb["AND"])))
for name, q in qs:
    row = []
    for fl, lab in ((A.COUNT, "default"), (A.COUNT | A.FORCE_FULLSCAN, "fullscan")):
        for _ in range(2):
            r = q.scan_device(t.data_ptr(), n, flags=fl)
        t0 = time.perf_counter()
        for _ in range(4):
            r = q.scan_device(t.data_ptr(), n, flags=fl)
        dt = (time.perf_counter() - t0) / 4
        row.append("%s %.3f ms %.0f GB/s matched %d engine %d" % (lab, dt * 1e3, n / 1e9 / dt, r.n_matched, r.engine))
    rn = q.scan_device(t.data_ptr(), n)
    print(name, q.info(), " | ".join(row), "| numbered matched", rn.n_matched, flush=True)
    q.close()
