"""Run by tests/test_gpu_records.py::test_shipped_defaults_in_a_fresh_process in a process without AGH_*
variables: the library as shipped (switches read once per query, two-kernel count-only form below 4 GiB, the fused
kernel from there on).  A seeded part of test_gpu_parity.py's cases, the CLI-sized file path, and one 4 GiB count:
fused kernel == two-kernel form == the planted records."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert not [k for k in os.environ if k.startswith("AGH_") and k != "AGH_REQUIRE_GPU"], "run me without AGH_* switches"

import numpy as np          # noqa: E402
import torch                # noqa: E402
import agrep_amd as A       # noqa: E402
import _oracle as O         # noqa: E402
from _cases import _rand_case   # noqa: E402

assert A.device_count() >= 1
n_cases = 0
# 1. the headline pattern, every k, both engines, records and counts
text, planted = O.corpus(512, seed=12345, variants=O.VARIANTS_C2, plant_period=50)
tb = text.tobytes()
for k in (0, 1, 2, 3):
    for nocase in (False, True):
        want = O.asearch(O.PATTERN_C2, k, tb, nocase=nocase, cap=200000)
        with A.Query(O.PATTERN_C2, k, nocase=nocase) as q:
            res, ms = q.scan_buffer(text, cap=200000)
            res_c, _ = q.scan_buffer(text, flags=A.COUNT)
            res_f, ms_f = q.scan_buffer(text, flags=A.FORCE_FULLSCAN, cap=200000)
            res_l, _ = q.scan_buffer(text, flags=A.FILENAMEONLY)
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, ("filter", k, nocase)
        assert (res_f.n_matched, [(s, e) for s, e, _ in ms_f]) == want, ("fullscan", k, nocase)
        assert res_c.n_matched == want[0] and res_c.fused_segments == 0, ("count-only: two kernels below 4 GiB", k, nocase)
        assert (res_l.n_matched > 0) == (want[0] > 0)
        n_cases += 4
# 2. seeded random cases (the generator of the fuzz tests)
rng = random.Random(20260926)
for _ in range(40):
    pat, k, txt = _rand_case(rng, rng.choice((2, 4, 27)))
    want = O.asearch(pat, k, txt, cap=200000)
    with A.Query(pat, k) as q:
        res, ms = q.scan_buffer(np.frombuffer(txt, dtype=np.uint8), cap=200000)
        res_c, _ = q.scan_buffer(np.frombuffer(txt, dtype=np.uint8), flags=A.COUNT)
    assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want and res_c.n_matched == want[0], (pat, k)
    n_cases += 2
# 3. the table engine and -f at their shipped switches
with A.Query.pattern(b"approx#match", 1) as q:
    res_t, _ = q.scan_buffer(text, flags=A.COUNT)
    res_tn, _ = q.scan_buffer(text, flags=A.COUNT | A.FORCE_NUMBERED)
assert res_t.n_matched == res_tn.n_matched > 0
pats = [b"approxim", b"atematch", b"zzzzqqqq"]
with A.Query.multi(pats) as q:
    res_m, _ = q.scan_buffer(text, flags=A.COUNT)
assert res_m.n_matched == O.multi_exact_count(pats, tb)[0]
n_cases += 3
# ... -f with one error: a selective set (k_mscan), a dense one (k_mtile, count-only and by record number) and a short word
# with errors on the fast full scan (three text streams per lane), each against the union of single-pattern oracle scans
for pats, k in (([b"approxim", b"atematch", b"zzzzqqqq"], 1), ([b"appr", b"match", b"atema", b"zqzq"], 1)):
    want_u = set()
    for p_ in pats:
        want_u.update(O.asearch(p_, k, tb, cap=400000)[1])
    with A.Query.multi(pats, k=k) as q:
        rc, _ = q.scan_buffer(text, flags=A.COUNT)
        rl, ms = q.scan_buffer(text, cap=400000)
    assert rc.fused_segments == 1 and rc.n_matched == len(want_u), (pats, rc.n_matched, len(want_u))
    assert [(s_, e_) for s_, e_, _ in ms] == sorted(want_u), pats
    n_cases += 2
with A.Query(b"matching", 2) as q:
    rw, msw = q.scan_buffer(text, cap=400000)
    assert rw.engine == A.ENGINE_FULLSCAN and (rw.n_matched, [(s_, e_) for s_, e_, _ in msw]) == O.asearch(b"matching", 2, tb, cap=400000)
n_cases += 1
# ... the table engine on records of ~1.7 KB (delimiter 's' + newline) and on a delimiter of two bytes: the fast kernels count
# pieces with one record end themselves and hand the stragglers of the walk to k_table_cont -- count, list and 4 GiB against
# the same scan with both switched off
for delim in (b"s\n", b"e "):
    tbl = A.compile_pattern(b"approx#match", delim=delim)
    ot = O.tables_from_golden({"Mask": list(tbl.Mask), "Init0": tbl.Init0, "Init1": tbl.Init1, "NO_ERR_MASK": tbl.NO_ERR_MASK,
                               "endposition": tbl.endposition, "D_endpos": tbl.D_endpos, "wildmask": tbl.wildmask,
                               "AND": tbl.AND}, tbl.M, dlen=len(delim))
    want_t = O.asearch_tables(ot, 1, tb, delim=delim, cap=400000)
    with A.Query.pattern(b"approx#match", 1, delim=delim) as q:
        rt, mst = q.scan_buffer(text, cap=400000)
        rtc, _ = q.scan_buffer(text, flags=A.COUNT)
    assert want_t[0] > 0 and (rt.n_matched, [(s_, e_) for s_, e_, _ in mst]) == want_t and rtc.n_matched == want_t[0], delim
    n_cases += 2
# 4. 4 GiB resident: the fused kernel (the shipped form from 4 GiB on) against two kernels and the planted records
n = 4 << 30
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
planted = A.corpus_fill_device(buf.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
want = sum(c for c, e in zip(planted, (0, 0, 1, 1, 2, 2, 2)) if e <= 2)
with A.Query(O.PATTERN_C2, 2) as q:
    r_fused = q.scan_device(buf.data_ptr(), n, flags=A.COUNT)
os.environ["AGH_FUSED"] = "0"
with A.Query(O.PATTERN_C2, 2) as q:
    r_two = q.scan_device(buf.data_ptr(), n, flags=A.COUNT)
del os.environ["AGH_FUSED"]
assert r_fused.fused_segments == 1 and r_two.fused_segments == 0
assert r_fused.n_matched == r_two.n_matched == want, (r_fused.n_matched, r_two.n_matched, want)
with A.Query.pattern(b"approx#match", 1, delim=b"s\n") as q:
    r_cont = q.scan_device(buf.data_ptr(), n, flags=A.COUNT)
os.environ["AGH_TF_CONT"] = "0"
os.environ["AGH_TF_DIRECT"] = "0"
with A.Query.pattern(b"approx#match", 1, delim=b"s\n") as q:
    r_walk = q.scan_device(buf.data_ptr(), n, flags=A.COUNT)
del os.environ["AGH_TF_CONT"], os.environ["AGH_TF_DIRECT"]
assert r_cont.n_matched == r_walk.n_matched > 1000, (r_cont.n_matched, r_walk.n_matched)
print("shipped defaults ok: %d cases, 4 GiB fused == two kernels == planted == %d; table engine, 1.7 KB records, 4 GiB: %d == %d"
      % (n_cases, want, r_cont.n_matched, r_walk.n_matched))
