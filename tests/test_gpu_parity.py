"""GPU parity tests: the HIP path, called through the C-ABI (libagrep_hip.so via ctypes),
against the CPU oracle on the same inputs.  Bit-exact: counts AND matched-record sets."""
import json
import os
import random

import numpy as np
import pytest

import _oracle as O
from _cases import _rand_case

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def agh():
    import agrep_amd
    assert agrep_amd.device_count() >= 1, "GPU tests need a HIP device"
    return agrep_amd


def _gpu(agh, pat, k, text, nocase=False, flags=0, cap=200000, delim=b"\n"):
    with agh.Query(pat, k, nocase=nocase, delim=delim) as q:
        res, ms = q.scan_buffer(text, flags=flags, cap=cap)
    assert not res.truncated
    return res, [(s, e) for s, e, _ in ms], [i for _, _, i in ms]


def _check(agh, pat, k, text, nocase=False):
    """Both device engines against orc_asearch (m <= 29) or the multi-word oracle."""
    tb = text.tobytes() if isinstance(text, np.ndarray) else bytes(text)
    if len(pat) <= 29:
        want = O.asearch(pat, k, tb, nocase=nocase, cap=200000)
    else:
        want = O.wm_count(pat, k, tb, nocase=nocase, word_bits=64, cap=200000)
    res_f, recs_f, idx_f = _gpu(agh, pat, k, text, nocase, agh.FORCE_FULLSCAN)
    assert res_f.engine == (agh.ENGINE_FULLSCAN if tb else 0)
    assert (res_f.n_matched, recs_f) == want, ("fullscan", pat, k, nocase)
    res, recs, idx = _gpu(agh, pat, k, text, nocase)
    assert (res.n_matched, recs) == want, ("default", pat, k, nocase, res.engine)
    assert idx == idx_f
    # count-only scans take the lean path (record = offset of its first byte, no census)
    with agh.Query(pat, k, nocase=nocase) as q:
        res_c, _ = q.scan_buffer(text, flags=agh.COUNT)
        res_n, _ = q.scan_buffer(text, flags=agh.COUNT | agh.FORCE_NUMBERED)
    assert res_c.n_matched == want[0], ("lean", pat, k, nocase)
    assert res_n.n_matched == want[0] and res_n.n_records == res.n_records
    # the lean path is one fused kernel where the query's shape has an instance; the two-kernel
    # form (k_sweep, then k_verify) serves the other shapes and -l: keep both honest
    os.environ["AGH_FUSED"] = "0"
    try:
        with agh.Query(pat, k, nocase=nocase) as q:
            res_2, _ = q.scan_buffer(text, flags=agh.COUNT)
    finally:
        del os.environ["AGH_FUSED"]
    assert res_2.n_matched == want[0], ("lean, two kernels", pat, k, nocase)
    # ... and so does the count-only full scan (no census pass in front of k_fullscan)
    with agh.Query(pat, k, nocase=nocase) as q:
        res_lf, _ = q.scan_buffer(text, flags=agh.COUNT | agh.FORCE_FULLSCAN)
    assert res_lf.n_matched == want[0], ("lean fullscan", pat, k, nocase)
    assert res_lf.engine == (agh.ENGINE_FULLSCAN if tb else 0)
    # record numbers are consistent with the record starts
    nl = np.frombuffer(tb, dtype=np.uint8) == 10
    for (s, e), i in zip(recs[:50], idx[:50]):
        assert int(nl[:s].sum()) == i
    n_rec = int(nl.sum()) + (1 if tb and tb[-1] != 10 else 0)
    assert res.n_records == n_rec == res_f.n_records
    return res


def test_device_corpus_generator_matches_cpu_twin(agh):
    import torch
    for upper in (0, 500):
        cpu, planted_cpu = O.corpus(64, first_page=7, seed=4242, variants=O.VARIANTS_C2,
                                    plant_period=25, upper_permille=upper)
        t = torch.empty(64 * 4096, dtype=torch.uint8, device="cuda")
        planted = agh.corpus_fill_device(t.data_ptr(), 64, first_page=7, seed=4242,
                                         variants=O.VARIANTS_C2, plant_period=25,
                                         upper_permille=upper)
        assert planted == planted_cpu
        assert np.array_equal(t.cpu().numpy(), cpu)


@pytest.mark.parametrize("k", [0, 1, 2, 3])
@pytest.mark.parametrize("nocase", [False, True])
def test_headline_pattern_on_generated_corpus(agh, k, nocase):
    """Config C2 shape (m=16, newline records) at a size the oracle finishes in seconds."""
    text, planted = O.corpus(512, seed=12345, variants=O.VARIANTS_C2, plant_period=40,
                             upper_permille=500 if nocase else 0)
    res = _check(agh, O.PATTERN_C2, k, text, nocase)
    # m=16: the sample lemma admits a filter up to k=2 (test_filter_shape_selection); k=3 goes
    # through the piece engine (4 pieces of 4 bytes), which is a filter as well
    assert res.engine == agh.ENGINE_FILTER
    if k >= 2 and not nocase:
        assert res.n_matched >= sum(planted)


def test_filter_shape_selection(agh):
    """floor((m-k-q+1)/h) >= k+1 (DESIGN.md 'sample lemma'); (16, 2) takes the H = 2 shape."""
    os.environ.pop("AGH_SHAPE_H2", None)
    want = {(16, 0): (4, 8), (16, 1): (4, 4), (16, 2): (4, 2), (48, 3): (4, 8), (8, 1): (0, 0),
            (8, 0): (4, 4), (29, 4): (4, 4), (29, 5): (0, 0), (64, 3): (4, 8), (4, 0): (0, 0)}
    for (m, k), (fq, fh) in want.items():
        with agh.Query(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-+"[:m], k) as q:
            info = q.info()
        assert (info["filter_q"], info["filter_h"]) == (fq, fh), (m, k, info)
        if fq and fh > 2:
            assert (m - k - fq + 1) // fh >= k + 1
        elif fq:                    # H = 2: the 4-byte samples overlap, one error spoils two
            assert (m - k - fq + 1) // fh >= 2 * k + 1


def _load_scan():
    with open(os.path.join(GOLD, "scan.json")) as f:
        return [c for c in json.load(f)["cases"] if not c.get("delim")]


def test_reference_golden_counts(agh):
    """Every newline-delimited fixture the reference produced (tests/golden/scan.json)."""
    n = 0
    for case in _load_scan():
        spec = case["text"]
        if spec["kind"] == "literal":
            text = spec["latin1"].encode("latin1")
        else:
            text = O.corpus(spec["pages"], seed=spec["seed"], variants=O.VARIANTS_C2,
                            plant_period=spec["period"])[0].tobytes()
        pat = case["pattern"].encode("latin1")
        nocase = "-i" in case["opts"]
        res, recs, _ = _gpu(agh, pat, case["k"], text, nocase)
        want = O.asearch(pat, case["k"], text, nocase=nocase, cap=200000)
        assert (res.n_matched, recs) == want
        sgrep_path = (not nocase) and case["k"] > 0
        if not sgrep_path:
            assert res.n_matched == case["count"]       # the reference's own number
        else:
            assert res.n_matched >= case["count"]       # Q2 only loses records
        n += 1
    assert n > 50


@pytest.mark.parametrize("sigma", [2, 4, 27])
def test_fuzz_small_texts(agh, sigma):
    rng = random.Random(2000 + sigma)
    for it in range(40):
        pat, k, text = _rand_case(rng, sigma)
        _check(agh, pat, k, text)


def test_fuzz_long_patterns_multiword(agh):
    """m in 30..64: 64-bit state words (config C3 shape m=48, k=3, -i)."""
    rng = random.Random(4848)
    for it in range(25):
        pat, k, text = _rand_case(rng, 27, max_m=64)
        k = min(k, 4)
        text = bytes(c - 32 if (97 <= c <= 122 and rng.random() < 0.5) else c for c in text)
        _check(agh, pat, k, text, nocase=True)


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 1023, 1024, 1025, 4095, 4096, 4097, 65535,
                               65536, 65537, 262143, 262144, 262145, 262144 * 3 + 5])
def test_sizes_around_strip_tile_and_wave_boundaries(agh, n):
    base, _ = O.corpus((n + 4095) // 4096 + 1, seed=n + 1, variants=O.VARIANTS_C2,
                       plant_period=7)
    _check(agh, O.PATTERN_C2, 2, base[:n].copy())
    _check(agh, O.PATTERN_C2, 0, base[:n].copy())


def test_edge_cases(agh):
    pat = b"needle"
    cases = [b"", b"\n", b"\n\n\n\n", b"needle", b"needle\n", b"\nneedle", b"x" * 100000 + b"needle",
             b"needle" + b"y" * 300000 + b"\nneedle\n", b"nee\ndle\n", b"neeedle\n" * 5000,
             (b"a" * 70 + b"\n") * 3000, b"\n".join([b"needle"] * 4000),
             b"needle " * 60000]
    for text in cases:
        for k in (0, 1, 2):
            _check(agh, pat, k, text)


def test_match_spanning_chunk_and_strip_boundaries(agh):
    """Occurrences placed across every kind of internal boundary of both kernels."""
    pat = O.PATTERN_C2
    for boundary in (256, 1024, 4096, 65536, 262144):
        for shift in (1, 5, 8, 15):
            n = boundary + 4096
            text = bytearray(b"z" * n)
            for i in range(80, n, 97):
                text[i] = 10
            at = boundary - shift
            text[at:at + 14] = b"apprximatemtch"          # 2 deletions
            for p in range(at - 3, at + 20):
                if text[p] == 10:
                    text[p] = ord("z")
            res = _check(agh, pat, 2, bytes(text))
            assert res.n_matched == 1


def test_from_maskgen_tables(agh):
    """The reference's own maskgen() output drives the device path (drop-in seam)."""
    with open(os.path.join(GOLD, "maskgen.json")) as f:
        cases = [c for c in json.load(f)["cases"] if not c.get("too_long") and "-d" not in c["opts"]]
    text, _ = O.corpus(64, seed=3, variants=O.VARIANTS_C2, plant_period=5, upper_permille=300)
    tb = text.tobytes()
    for c in cases:
        t = c["tables"]
        pat = c["pattern"].encode()
        M = len(pat) + 2
        q = agh.Query.from_maskgen(t["Mask"], t["Init0"], t["Init1"], t["NO_ERR_MASK"],
                                   t["endposition"], t["D_endpos"], M, b"\n", c["k"], t["AND"])
        res, ms = q.scan_buffer(tb, cap=100000)
        q.close()
        want = O.asearch(pat, c["k"], tb, nocase="-i" in c["opts"], cap=100000)
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, c["pattern"]


def test_scan_fd(agh, tmp_path):
    text, _ = O.corpus(40, seed=11, variants=O.VARIANTS_C2, plant_period=9)
    p = tmp_path / "corpus.txt"
    p.write_bytes(text.tobytes())
    fd = os.open(str(p), os.O_RDONLY)
    try:
        with agh.Query(O.PATTERN_C2, 2) as q:
            res, ms = q.scan_fd(fd, cap=10000)
    finally:
        os.close(fd)
    assert (res.n_matched, [(s, e) for s, e, _ in ms]) == O.asearch(O.PATTERN_C2, 2, text.tobytes(), cap=10000)


def test_resident_corpus_properties_at_scale(agh):
    """1 GiB resident in HBM: engines agree, counts are monotone in k, planted records found,
    and a 16 MiB slice agrees with the oracle."""
    import torch
    pages = (1 << 30) // 4096
    t = torch.empty(pages * 4096, dtype=torch.uint8, device="cuda")
    planted = agh.corpus_fill_device(t.data_ptr(), pages, seed=12345, variants=O.VARIANTS_C2,
                                     plant_period=500)
    by_edits = [planted[0] + planted[1], planted[2] + planted[3], sum(planted[4:7])]
    prev = 0
    for k in (0, 1, 2, 3):
        with agh.Query(O.PATTERN_C2, k) as q:
            r1 = q.scan_device(t.data_ptr(), t.numel())
            r2 = q.scan_device(t.data_ptr(), t.numel(), flags=agh.FORCE_FULLSCAN)
            r3 = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT)
        assert r3.n_matched == r1.n_matched
        assert r1.engine == agh.ENGINE_FILTER       # k = 3: the piece engine
        assert r2.engine == agh.ENGINE_FULLSCAN
        assert r1.n_matched == r2.n_matched and r1.n_records == r2.n_records
        assert r1.n_matched >= prev
        assert r1.n_matched >= sum(by_edits[:min(k, 2) + 1])
        prev = r1.n_matched
    sl = t[:16 << 20].cpu().numpy()
    with agh.Query(O.PATTERN_C2, 2) as q:
        r = q.scan_device(t.data_ptr(), 16 << 20)
    assert r.n_matched == O.asearch(O.PATTERN_C2, 2, sl)[0]


def _c3_pattern_and_variants():
    rng = random.Random(48)
    pat = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(48))
    vs = [pat]
    for edits in (1, 2, 3, 4):
        v = bytearray(pat)
        for _ in range(edits):
            op, pos = rng.randint(0, 2), rng.randrange(4, len(v) - 4)
            if op == 0:
                v[pos] = ord("Q")
            elif op == 1:
                del v[pos]
            else:
                v.insert(pos, ord("Z"))
        vs.append(bytes(v))
    return pat, tuple(vs)


def test_config_c3_shape_long_pattern_nocase(agh):
    """BASELINE config 3 shape: m=48 (64-bit state words), k=3, -i, mixed-case text.
    The reference rejects m > 29, so the anchor is the multi-word oracle / DP."""
    import torch
    pat, vs = _c3_pattern_and_variants()
    text, planted = O.corpus(768, seed=3, variants=vs, plant_period=25, upper_permille=500)
    res = _check(agh, pat, 3, text, nocase=True)
    assert res.engine == agh.ENGINE_FILTER
    assert res.n_matched >= sum(planted[:4])          # 0..3 edits must be found
    # at scale: engines agree and lean == numbered
    pages = (1 << 30) // 4096
    t = torch.empty(pages * 4096, dtype=torch.uint8, device="cuda")
    planted = agh.corpus_fill_device(t.data_ptr(), pages, seed=9, variants=vs, plant_period=500,
                                     upper_permille=500)
    with agh.Query(pat, 3, nocase=True) as q:
        a = q.scan_device(t.data_ptr(), t.numel())
        b = q.scan_device(t.data_ptr(), t.numel(), flags=agh.FORCE_FULLSCAN)
        c = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT)
    assert a.n_matched == b.n_matched == c.n_matched >= sum(planted[:4])
    assert a.n_records == b.n_records


def test_segmented_scan_cuts_at_record_boundaries(agh, monkeypatch):
    """Inputs above the segment limit are cut at (16-byte aligned) record boundaries and the
    per-segment results add up to the single-segment answer."""
    import torch
    pages = (96 << 20) // 4096
    t = torch.empty(pages * 4096, dtype=torch.uint8, device="cuda")
    agh.corpus_fill_device(t.data_ptr(), pages, seed=21, variants=O.VARIANTS_C2, plant_period=60)
    with agh.Query(O.PATTERN_C2, 2) as q:
        whole = q.scan_device(t.data_ptr(), t.numel())
        whole_lean = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT)
    monkeypatch.setenv("AGH_SEG_MAX_MB", "20")
    with agh.Query(O.PATTERN_C2, 2) as q:
        parts = q.scan_device(t.data_ptr(), t.numel())
        parts_lean = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT)
        parts_full = q.scan_device(t.data_ptr(), t.numel(), flags=agh.FORCE_FULLSCAN)
    assert parts.n_matched == whole.n_matched == whole_lean.n_matched == parts_lean.n_matched
    assert parts_full.n_matched == whole.n_matched
    assert parts.n_records == whole.n_records == parts_full.n_records
    # record lists of a segmented scan: positions and record numbers are absolute
    # (regression: they were relative to the segment that found them)
    tb = t[:48 << 20].cpu().numpy().tobytes()
    want = O.asearch(O.PATTERN_C2, 2, tb, cap=400000)
    with agh.Query(O.PATTERN_C2, 2) as q:
        res, ms = q.scan_buffer(tb, cap=400000)           # 48 MiB in 20 MiB segments
        inv, ms_inv = q.scan_buffer(tb[:30 << 20], flags=agh.INVERT, cap=600000)
    assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want
    nl = np.flatnonzero(np.frombuffer(tb, dtype=np.uint8) == 10)
    for s, e, i in ms[:20] + ms[-20:]:
        assert int(np.searchsorted(nl, s)) == i
    assert inv.n_matched == len(ms_inv) == inv.n_records - sum(1 for s, e, _ in ms if e <= (30 << 20))
    assert all(a[0] < b[0] and a[2] + 1 <= b[2] for a, b in zip(ms_inv, ms_inv[1:]))


def test_scan_fd_streams_files_and_pipes(agh, tmp_path):
    """agh_scan_fd: pinned double-buffered staging (several 32 MiB chunks), regular file and
    pipe; matched records come back through the device-side gather (agh_fetch_records)."""
    import threading
    text, _ = O.corpus((80 << 20) // 4096, seed=31, variants=O.VARIANTS_C2, plant_period=3000)
    tb = text.tobytes()
    want_n, want_recs = O.asearch(O.PATTERN_C2, 2, tb, cap=100000)
    p = tmp_path / "big.txt"
    p.write_bytes(tb)
    with agh.Query(O.PATTERN_C2, 2) as q:
        fd = os.open(str(p), os.O_RDONLY)
        try:
            res, ms = q.scan_fd(fd, cap=100000)
        finally:
            os.close(fd)
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == (want_n, want_recs)
        recs = q.fetch_records(ms)
        assert recs == [tb[s:e] for s, e in want_recs]
        # too small a match array: truncated, then the staged text is scanned again
        res_t, ms_t = q.scan_fd(os.open(str(p), os.O_RDONLY), cap=5)
        assert res_t.truncated and res_t.n_matched == want_n and len(ms_t) == 5
        # a pipe: length unknown in advance, the device buffer grows
        r, w = os.pipe()

        def feed():
            with os.fdopen(w, "wb") as f:
                for i in range(0, len(tb), 1 << 20):
                    f.write(tb[i:i + (1 << 20)])
        th = threading.Thread(target=feed)
        th.start()
        try:
            res_p, ms_p = q.scan_fd(r, cap=100000)
        finally:
            th.join()
            os.close(r)
        assert (res_p.n_matched, [(s, e) for s, e, _ in ms_p]) == (want_n, want_recs)
        res_c, _ = q.scan_fd(os.open(str(p), os.O_RDONLY), flags=agh.COUNT)
        assert res_c.n_matched == want_n


def _emit_all(batches):
    ms = [m for b in batches for m in b[0]]
    recs = [r for b in batches for r in (b[1] or [])]
    return ms, recs


@pytest.mark.parametrize("seg_mb", ["1024", "16"])
def test_scan_fd_emit_streams_records(agh, tmp_path, monkeypatch, seg_mb):
    """agh_scan_fd_emit: the input streams through two bounded device segments (16 MiB here: six segments,
    the residue carried five times), matched records come out per segment in file order -- the same
    offsets, record numbers and bytes as the oracle / the whole-file path, from a file and from a pipe;
    stopping from inside emit(); offsets only (AGH_NO_BYTES); -v; a byte range."""
    import threading
    monkeypatch.setenv("AGH_STREAM_SEG_MB", seg_mb)
    text, _ = O.corpus((80 << 20) // 4096, seed=31, variants=O.VARIANTS_C2, plant_period=3000)
    tb = text.tobytes()
    want_n, want_recs = O.asearch(O.PATTERN_C2, 2, tb, cap=100000)
    nl = np.flatnonzero(text == 10)
    want_idx = np.searchsorted(nl, [s for s, _ in want_recs]).tolist()
    p = tmp_path / "big.txt"
    p.write_bytes(tb)
    with agh.Query(O.PATTERN_C2, 2) as q:
        fd = os.open(str(p), os.O_RDONLY)
        try:
            res, batches = q.scan_fd_emit(fd)
        finally:
            os.close(fd)
        ms, recs = _emit_all(batches)
        assert res.n_matched == want_n and not res.truncated
        assert [(s, e) for s, e, _ in ms] == want_recs and [i for _, _, i in ms] == want_idx
        assert recs == [tb[s:e] for s, e in want_recs]
        assert len(batches) >= (5 if seg_mb == "16" else 1)
        assert res.n_bytes == len(tb) and res.n_records == len(nl)
        # a pipe
        r, w = os.pipe()

        def feed():
            with os.fdopen(w, "wb") as f:
                for i in range(0, len(tb), 1 << 20):
                    f.write(tb[i:i + (1 << 20)])
        th = threading.Thread(target=feed)
        th.start()
        try:
            res_p, bp = q.scan_fd_emit(r)
        finally:
            th.join()
            os.close(r)
        assert _emit_all(bp)[0] == ms and _emit_all(bp)[1] == recs
        # offsets only; stop after the first batch
        fd = os.open(str(p), os.O_RDONLY)
        try:
            res_o, bo = q.scan_fd_emit(fd, flags=agh.NO_BYTES)
            os.lseek(fd, 0, os.SEEK_SET)
            res_s, bs = q.scan_fd_emit(fd, stop_after=1)
            os.lseek(fd, 0, os.SEEK_SET)
            half = int(nl[len(nl) // 2]) + 1                  # a record-aligned byte range
            res_r, br = q.scan_fd_emit(fd, byte_range=(half, len(tb)))
        finally:
            os.close(fd)
        assert _emit_all(bo)[0] == ms and _emit_all(bo)[1] == []
        assert len(bs) == 1 and (res_s.truncated == 1 or len(batches) == 1)
        assert [(s + half, e + half) for s, e, _ in _emit_all(br)[0]] == [x for x in want_recs if x[0] >= half]
        # count-only scans of the same file take the same pipeline
        fd = os.open(str(p), os.O_RDONLY)
        try:
            assert q.scan_fd(fd, flags=agh.COUNT)[0].n_matched == want_n
        finally:
            os.close(fd)
    # -v on a small file in many segments: every record that does not match
    small = tb[:3 << 20]
    ps = tmp_path / "small.txt"
    ps.write_bytes(small)
    monkeypatch.setenv("AGH_STREAM_SEG_MB", "1")
    got_n, got = O.asearch(O.PATTERN_C2, 2, small, cap=100000)
    all_recs, pos = [], 0
    for e in np.flatnonzero(np.frombuffer(small, dtype=np.uint8) == 10).tolist():
        all_recs.append((pos, e))
        pos = e + 1
    want_inv = [x for x in all_recs if x not in set(got)]
    with agh.Query(O.PATTERN_C2, 2) as q:
        fd = os.open(str(ps), os.O_RDONLY)
        try:
            res_v, bv = q.scan_fd_emit(fd, flags=agh.INVERT)
        finally:
            os.close(fd)
    assert [(s, e) for s, e, _ in _emit_all(bv)[0]] == want_inv and res_v.n_matched == len(want_inv)


@pytest.mark.parametrize("delim", [b"\r\n", b"\n\n", b"aa", b"From "])
def test_scan_fd_emit_multi_byte_delimiters(agh, tmp_path, monkeypatch, delim):
    """The streaming pipeline with delimiters of several bytes (also ones that overlap themselves: a
    segment is cut only behind an occurrence no other occurrence overlaps from the left), count-only and
    with records, 1 MiB segments."""
    monkeypatch.setenv("AGH_STREAM_SEG_MB", "1")
    rng = random.Random(len(delim) + delim[0])
    text, _ = O.corpus((6 << 20) // 4096, seed=77, variants=O.VARIANTS_C2, plant_period=200)
    tb = text.tobytes().replace(b"\n", delim)
    if delim == b"\n\n":                                   # runs of the delimiter's own byte
        tb = tb.replace(b" e", b"\n\n\ne", 2000)
    want_n, want_recs = O.asearch(O.PATTERN_C2, 2, tb, delim=delim, cap=200000)
    p = tmp_path / "mb.txt"
    p.write_bytes(tb)
    with agh.Query(O.PATTERN_C2, 2, delim=delim) as q:
        fd = os.open(str(p), os.O_RDONLY)
        try:
            rc, _ = q.scan_fd(fd, flags=agh.COUNT)
            os.lseek(fd, 0, os.SEEK_SET)
            res, batches = q.scan_fd_emit(fd)
        finally:
            os.close(fd)
    ms, recs = _emit_all(batches)
    assert rc.n_matched == want_n == res.n_matched > 10
    assert [(s, e) for s, e, _ in ms] == want_recs
    assert recs == [tb[s:e] for s, e in want_recs]
    assert len(batches) >= 3


@pytest.mark.parametrize("delim", [b"FROM ", b"\n\n", b"#%", b"@@@", b"ab", b"aa", b"\n.\n", b"<record>", b"ENDREC\n", b"abcabc",
                                   b"aaaaaaaa"])
def test_multi_byte_delimiters(agh, delim):
    """-d with 2..8 byte delimiters (-d 'From ', -d '$$'): leftmost non-overlapping delimiter
    occurrences, reset after the delimiter's last byte -- bit-exact with asearch.c, including
    delimiters that overlap themselves ("\\n\\n", "aa") or share letters with the pattern."""
    rng = random.Random(len(delim) * 131 + delim[0])
    for it in range(30):
        pat, k, text = _rand_case(rng, 6, max_m=20)
        text = text.replace(b"\n", delim if rng.random() < 0.7 else b"\n")
        if it % 5 == 0:
            text = text + delim * rng.randint(1, 4)            # runs of delimiters at the end
        if it % 7 == 0:
            text = delim * rng.randint(1, 5) + text
        want = O.asearch(pat, k, text, delim=delim, cap=100000)
        for flags in (0, agh.FORCE_FULLSCAN):
            res, recs, _ = _gpu(agh, pat, k, text, flags=flags, delim=delim)
            assert (res.n_matched, recs) == want, (delim, pat, k, flags, text)
        with agh.Query(pat, k, delim=delim) as q:
            rc, _ = q.scan_buffer(text, flags=agh.COUNT)
        assert rc.n_matched == want[0], ("lean", delim, pat, k, text)
    # a planted corpus with paragraph / mbox style delimiters, all engines
    base, _ = O.corpus(64, seed=5, variants=O.VARIANTS_C2, plant_period=6)
    text = base.tobytes().replace(b"\n", delim)
    for k in (0, 2, 3):
        want = O.asearch(O.PATTERN_C2, k, text, delim=delim, cap=100000)
        for flags in (0, agh.FORCE_FULLSCAN):
            res, recs, _ = _gpu(agh, O.PATTERN_C2, k, text, flags=flags, delim=delim)
            assert (res.n_matched, recs) == want, (delim, k, flags)
        with agh.Query(O.PATTERN_C2, k, delim=delim) as q:
            assert q.scan_buffer(text, flags=agh.COUNT)[0].n_matched == want[0]


def test_filter_with_delimiter_bytes_inside_the_pattern(agh, monkeypatch):
    """Round 5: under a delimiter of several bytes a pattern position that accepts one of the delimiter's bytes no
    longer ends the sampled run (`-d 'From '` against any pattern with an r, o or m) -- only the positions a byte
    OUTSIDE the text can reach do: the last dlen + k ones (the delimiter appended at the end) and the first k + 1 (the
    byte in front of a segment).  Delimiters cut out of the pattern itself, occurrences that run into a delimiter at a
    record's end, start on its last byte, or end inside the appended one; default engine (the filter where a shape
    exists), full scan, count-only and numbered count against asearch.c -- also with 1 MiB segments (the byte in front
    of a segment is the previous delimiter's last)."""
    # (AGH_FUZZ_SEED: another walk; profiles/r05_fuzz_delim_bytes.log holds 40 of them)
    rng = random.Random(int(os.environ.get("AGH_FUZZ_SEED", "2026")))
    filtered = 0
    for it in range(120):
        sigma = rng.choice((4, 6, 8))
        alpha = bytes(rng.sample(range(97, 123), sigma - 1)) + b" "
        m = rng.randint(8, 24)
        k = rng.randint(0, 2)
        pat = bytes(rng.choice(alpha[:-1]) for _ in range(m))
        dlen = rng.randint(2, 5)
        at = rng.randint(0, m - dlen)
        delim = pat[at:at + dlen] if rng.random() < 0.7 else bytes(rng.choice(alpha) for _ in range(dlen))
        recs = []
        for _ in range(rng.randint(1, 80)):
            L = rng.randint(0, 150)
            r = bytearray(rng.choice(alpha) for _ in range(L))
            if rng.random() < 0.5:
                v = bytearray(pat)
                for _ in range(rng.randint(0, k + 1)):
                    op, pos = rng.randint(0, 2), rng.randrange(len(v))
                    if op == 0:
                        v[pos] = rng.choice(alpha)
                    elif op == 1 and len(v) > 1:
                        del v[pos]
                    else:
                        v.insert(pos, rng.choice(alpha))
                where = rng.random()
                if where < 0.3:                  # at the record's end: the tail of the copy runs into the delimiter
                    r += v[:rng.randint(max(1, len(v) - dlen - 1), len(v))]
                elif where < 0.5:                # at the record's start, minus what the previous delimiter may supply
                    r = v[rng.randint(0, 2):] + r
                elif L > len(v):
                    o = rng.randint(0, L - len(v))
                    r[o:o + len(v)] = v
            recs.append(bytes(r))
        text = delim.join(recs)
        tail = rng.random()
        if tail < 0.4:
            text += delim
        elif tail < 0.6:
            text += delim[:rng.randint(1, dlen - 1)]        # a delimiter cut short by the end of the text
        if it % 9 == 0 and delim in text:
            text = text * (1 + (3 << 20) // max(1, len(text)))      # > 3 MiB: several segments below
            monkeypatch.setenv("AGH_SEG_MAX_MB", "1")
        else:
            monkeypatch.delenv("AGH_SEG_MAX_MB", raising=False)
        want = O.asearch(pat, k, text, delim=delim, cap=400000)
        with agh.Query(pat, k, delim=delim) as q:
            res, ms = q.scan_buffer(text, cap=400000)
            res_f, ms_f = q.scan_buffer(text, flags=agh.FORCE_FULLSCAN, cap=400000)
            res_c, _ = q.scan_buffer(text, flags=agh.COUNT)
            res_n, _ = q.scan_buffer(text, flags=agh.COUNT | agh.FORCE_NUMBERED)
        assert (res_f.n_matched, [(s, e) for s, e, _ in ms_f]) == want, ("full scan", pat, k, delim, it)
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, ("default", res.engine, pat, k, delim, it)
        assert res_c.n_matched == res_n.n_matched == want[0], ("counts", pat, k, delim, it)
        filtered += res.engine == agh.ENGINE_FILTER and any(c in delim for c in pat)
    assert filtered >= 20       # the filter did take patterns that hold delimiter bytes


def _golden(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)["cases"]


def test_weighted_costs(agh):
    """-I# -S# -D# (asearch1.c) on the general automaton: the reference's own numbers
    (tests/golden/costs.json) and the cost oracle on fuzzed inputs."""
    for case in _golden("costs.json"):
        spec = case["text"]
        text = spec["latin1"].encode("latin1") if spec["kind"] == "literal" else \
            O.corpus(spec["pages"], seed=spec["seed"], variants=O.VARIANTS_C2,
                     plant_period=spec["period"])[0].tobytes()
        pat = case["pattern"].encode()
        with agh.Query(pat, case["k"]) as q:
            q.set_costs(*case["costs"])
            res, ms = q.scan_buffer(text, cap=100000)
            res_c, _ = q.scan_buffer(text, flags=agh.COUNT)
        assert res.n_matched == case["count"] == res_c.n_matched, case["costs"]
        want = O.asearch_costs(pat, case["k"], tuple(case["costs"]), text, cap=100000)
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want
    rng = random.Random(11)
    for it in range(40):
        pat, k, text = _rand_case(rng, 5, max_m=24)
        if k == 0:
            continue
        costs = (rng.randint(1, 3), rng.randint(1, 3), rng.randint(1, 3))
        want = O.asearch_costs(pat, k, costs, text, cap=100000)
        with agh.Query(pat, k) as q:
            q.set_costs(*costs)
            res, ms = q.scan_buffer(text, cap=100000)
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (pat, k, costs)


def test_exact_segments_from_maskgen_tables(agh):
    """<...> segments arrive through maskgen's NO_ERR_MASK (agh_query_from_maskgen)."""
    for case in _golden("exact_segments.json"):
        spec = case["text"]
        text = O.corpus(spec["pages"], seed=spec["seed"], variants=O.VARIANTS_C2,
                        plant_period=spec["period"])[0].tobytes()
        t = case["tables"]
        m = len(case["pattern"]) - 2
        q = agh.Query.from_maskgen(t["Mask"], t["Init0"], t["Init1"], t["NO_ERR_MASK"],
                                   t["endposition"], t["D_endpos"], m + 2, b"\n", case["k"], t["AND"])
        res, ms = q.scan_buffer(text, cap=100000)
        q.close()
        assert res.n_matched == case["count"], case["pattern"]      # the reference's own count
        want = O.asearch_tables(O.tables_from_golden(t, m + 2), case["k"], text, cap=100000)
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want


def test_pattern_language_from_maskgen_tables(agh):
    """[classes], -w and -x handed over as the reference's own maskgen tables."""
    for case in _golden("pattern_language.json"):
        spec = case["text"]
        text = O.corpus(spec["pages"], seed=spec["seed"], variants=O.VARIANTS_C2,
                        plant_period=spec["period"])[0].tobytes() + case["extra_latin1"].encode("latin1")
        t = case["tables"]
        M = t["D_endpos"].bit_length()
        q = agh.Query.from_maskgen(t["Mask"], t["Init0"], t["Init1"], t["NO_ERR_MASK"],
                                   t["endposition"], t["D_endpos"], M, b"\n", case["k"], t["AND"])
        for flags in (0, agh.FORCE_FULLSCAN):
            res, ms = q.scan_buffer(text, cap=100000, flags=flags)
            assert res.n_matched == case["count"], (case["pattern"], case["opts"], flags)
            want = O.asearch_tables(O.tables_from_golden(t, M), case["k"], text, cap=100000)
            assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (case["pattern"], case["opts"])
        q.close()


def test_query_pattern_compiles_what_the_reference_compiles(agh):
    """agh_query_pattern: the pattern STRING and the options go in, the library compiles them itself
    (agh_pattern.cpp) -- the reference's own counts on its golden texts and the oracle's match list on the
    reference's tables, for classes, -w / -x, '#', ';' / ',' lists and ^ $ anchors, under newline and under
    delimiters of several bytes."""
    from test_oracle_golden import lang_delims_text
    for case in _golden("pattern_language.json"):
        spec = case["text"]
        text = O.corpus(spec["pages"], seed=spec["seed"], variants=O.VARIANTS_C2,
                        plant_period=spec["period"])[0].tobytes() + case["extra_latin1"].encode("latin1")
        t = case["tables"]
        M = t["D_endpos"].bit_length()
        q = agh.Query.pattern(case["pattern"].encode("latin1"), case["k"], word="-w" in case["opts"],
                              wholeline="-x" in case["opts"])
        res, ms = q.scan_buffer(text, cap=100000)
        q.close()
        assert res.n_matched == case["count"], (case["pattern"], case["opts"])
        want = O.asearch_tables(O.tables_from_golden(t, M), case["k"], text, cap=100000)
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (case["pattern"], case["opts"])
    for case in _golden("pattern_language_delims.json"):
        text, delim = lang_delims_text(case)
        opt = case["opts"][case["opts"].index("-d") + 1].encode("latin1").replace(b"$", b"\n")
        q = agh.Query.pattern(case["pattern"].encode("latin1"), case["k"], nocase=case["nocase"], delim=opt)
        res, _ = q.scan_buffer(text, flags=agh.COUNT)
        q.close()
        assert res.n_matched == case["count"], (case["pattern"], case["opts"])


def test_table_engine_two_streams_per_lane(agh, monkeypatch):
    """k_tablescan_fast2 (tables of M <= 15 positions: two chunks share every state word) against the one-stream
    kernel, the exact one-kernel form and the oracle on the reference's tables: record lengths from 0 to beyond a
    chunk, texts that end inside a tile pair / exactly on a tile / in the odd tile, matches at the last byte, the
    whole list and count-only."""
    rng = np.random.default_rng(2026)
    words = [b"car", b"cars", b"red", b"fast", b"scar", b"cat", b"approx", b"match", b"approxQmatch", b"mat", b" ", b" ", b"\n"]
    base = b"".join(words[i] for i in rng.integers(0, len(words), 420000))           # ~1.6 MiB
    long_rec = bytes(rng.integers(97, 123, 9000, dtype=np.uint8))                   # no newline: beyond a 4 KiB chunk
    tile = 64 * 4096
    texts = [base, base[:tile], base[:tile + 1], base[:2 * tile], base[:2 * tile + 17], base[:3 * tile - 5] + b"scar",
             base[:65536], base[:65536 + 1], base[:3 * 65536 - 5] + b"scar", base[:128 * 1024 + 17],
             base[:5000] + long_rec + b"cat\n" + long_rec + b"approx match" + long_rec, b"scar", b"", b"\n\n",
             base[:tile - 3] + b"cat", (b"x" * 4095 + b"\n") * 70 + b"red car"]
    for pat, k in ((b"approx#match", 0), (b"approx#match", 1), (b"appr#mat#ch", 2), (b"scar,cat", 0), (b"car;red", 1),
                   (b"cars;fast", 0), (b"ma#ch,c#t", 1), (b"a#h", 0)):
        tb = agh.compile_pattern(pat)
        assert tb.M <= 15
        ot = O.tables_from_golden({"Mask": list(tb.Mask), "Init0": tb.Init0, "Init1": tb.Init1, "NO_ERR_MASK": tb.NO_ERR_MASK,
                                   "endposition": tb.endposition, "D_endpos": tb.D_endpos, "wildmask": tb.wildmask, "AND": tb.AND}, tb.M)
        with agh.Query.pattern(pat, k) as q:
            for i, text in enumerate(texts):
                want = O.asearch_tables(ot, k, text, cap=400000)
                # (bytes per lane of the fast form: 4 KiB, 2 KiB, 1 KiB, by the size of the text)
                for env in ({"AGH_TF_CHUNK": "4096"}, {"AGH_TF_CHUNK": "2048"}, {"AGH_TF_CHUNK": "1024"}, {"AGH_TF_CHUNK": "0"},
                            {"AGH_TF_PACK2": "0", "AGH_TF_CHUNK": "4096"}, {"AGH_TF_PACK2": "0", "AGH_TF_CHUNK": "1024"},
                            {"AGH_FS_FAST": "0"}):
                    for key in ("AGH_TF_PACK2", "AGH_FS_FAST", "AGH_TF_CHUNK"):
                        monkeypatch.delenv(key, raising=False)
                    for key, v in env.items():
                        monkeypatch.setenv(key, v)
                    res, ms = q.scan_buffer(text, cap=400000)
                    assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (pat, k, i, env)
                    res_c, _ = q.scan_buffer(text, flags=agh.COUNT)
                    assert res_c.n_matched == want[0], (pat, k, i, env, "count-only")
    for key in ("AGH_TF_PACK2", "AGH_FS_FAST", "AGH_TF_CHUNK"):
        monkeypatch.delenv(key, raising=False)


def test_table_engine_adversarial_texts(agh):
    """'#' wildcards, ';' AND and ',' OR run by the table engine (agh_table.hip): records from
    empty to 300 KiB, chunk-straddling records, no trailing delimiter -- always the oracle's
    answer on the reference's own tables (which reproduced the reference counts on CPU)."""
    rng = np.random.default_rng(7)
    alpha = np.frombuffer(b"acrest\n", dtype=np.uint8)
    texts = [b"", b"car", b"\n", b"\n\n\n", b"cars are fast", b"red car\n\ncar red\nfast cars"]
    t = alpha[rng.integers(0, len(alpha), 200000)].tobytes()
    texts += [t, t.rstrip(b"\n") + b"s"]
    long_rec = alpha[rng.integers(0, len(alpha) - 1, 300 * 1024)].tobytes()      # no newline
    texts.append(b"red\n" + long_rec + b"car\n" + long_rec[:5000] + b"\nfast cars\n")
    words = [b"car", b"cars", b"red", b"fast", b"scar", b"cat", b"a", b"h", b"ah", b"arch", b" ", b"\n", b"\n"]
    texts.append(b"".join(words[i] for i in rng.integers(0, len(words), 60000)))
    for case in _golden("pattern_language.json"):
        tb = case["tables"]
        M = tb["D_endpos"].bit_length()
        q = agh.Query.from_maskgen(tb["Mask"], tb["Init0"], tb["Init1"], tb["NO_ERR_MASK"],
                                   tb["endposition"], tb["D_endpos"], M, b"\n", case["k"], tb["AND"])
        ot = O.tables_from_golden(tb, M)
        for i, text in enumerate(texts):
            res, ms = q.scan_buffer(text, cap=200000)
            want = O.asearch_tables(ot, case["k"], text, cap=200000)
            assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (case["pattern"], case["opts"], i)
            res_c, _ = q.scan_buffer(text, flags=agh.COUNT)
            assert res_c.n_matched == want[0]
        q.close()


def test_table_engine_with_multi_byte_delimiters(agh):
    """'#', ';' and ',' under delimiters of several bytes and letters under -i: record ends from the
    delimiter bitmap, the reference's tables run literally (agh_table.hip).  The reference's own counts
    on its golden texts, and the oracle on texts built to hurt: delimiters across strip / tile borders,
    runs of a delimiter that overlaps itself, no delimiter at the end, nothing but delimiters."""
    from test_oracle_golden import lang_delims_text
    rng = np.random.default_rng(23)
    for ci, case in enumerate(_golden("pattern_language_delims.json")):
        text, delim = lang_delims_text(case)
        tb = case["tables"]
        M = tb["D_endpos"].bit_length() + len(delim) - 1        # D_endpos = the LAST delimiter position, bit M - D_length
        opt = case["opts"][case["opts"].index("-d") + 1].encode("latin1")
        q = agh.Query.from_maskgen(tb["Mask"], tb["Init0"], tb["Init1"], tb["NO_ERR_MASK"], tb["endposition"],
                                   tb["D_endpos"], M, opt, case["k"], tb["AND"])
        ot = O.tables_from_golden(tb, M, dlen=len(delim))
        texts = [text]
        if ci % 8 in (0, 4, 5):                 # one '#', one ';' and one ',' pattern per delimiter
            letters = np.frombuffer(b"acrestpoxim h", dtype=np.uint8)
            body = letters[rng.integers(0, len(letters), 150000)].tobytes()
            pieces, pos = [], 0
            while pos < len(body):
                step = int(rng.integers(1, 400))
                pieces.append(body[pos:pos + step])
                pos += step
            joined = delim.join(pieces)
            # delimiters moved onto the 1 KiB / 64 KiB borders, a run of delimiter bytes, both ends
            b = bytearray(joined)
            for border in (1024, 4096, 65536, 131072):
                for shift in range(len(delim) + 1):
                    at = border - shift + 7 * 1024 * shift
                    if at + len(delim) < len(b):
                        b[at:at + len(delim)] = delim
            run_of = (delim[:1] * (2 * len(delim) + 1))
            b[70000:70000 + len(run_of)] = run_of
            texts += [bytes(b), delim + bytes(b[:5000]), bytes(b[:3000]) + b"approxQmatch cars fast scar",
                      delim * 5, b"car", b""]
        for ti, t in enumerate(texts):
            if not t and len(delim) > 1:
                continue                        # (empty text: only the appended delimiter, Q11)
            want = O.asearch_tables(ot, case["k"], t, delim=delim, cap=300000)
            if ti == 0:
                assert want[0] == case["count"]
            res, ms = q.scan_buffer(t, cap=300000)
            assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (case["pattern"], case["opts"], ti)
            res_c, _ = q.scan_buffer(t, flags=agh.COUNT)
            res_n, _ = q.scan_buffer(t, flags=agh.COUNT | agh.FORCE_NUMBERED)
            assert res_c.n_matched == res_n.n_matched == want[0], (case["pattern"], case["opts"], ti)
            # the same on the exact one-kernel form (the default above is the fast form + replay since round 5),
            # and with one stream per lane
            for sw in ("AGH_FS_FAST", "AGH_TF_PACK2"):
                os.environ[sw] = "0"
                try:
                    res_x, ms_x = q.scan_buffer(t, cap=300000)
                    res_xc, _ = q.scan_buffer(t, flags=agh.COUNT)
                finally:
                    del os.environ[sw]
                assert (res_x.n_matched, [(s, e) for s, e, _ in ms_x]) == want and res_xc.n_matched == want[0], (sw, case["pattern"], ti)
            # ... and with edit costs (asearch1.c's levels, fast form and exact kernel)
            if case["k"] >= 1 and ti <= 1:
                for costs in ((2, 1, 1), (1, 2, 2)):
                    want_c = O.asearch_tables_costs(ot, case["k"], costs, t, delim=delim, cap=300000)
                    q.set_costs(*costs)
                    try:
                        res_k, ms_k = q.scan_buffer(t, cap=300000)
                        res_kc, _ = q.scan_buffer(t, flags=agh.COUNT)
                        os.environ["AGH_FS_FAST"] = "0"
                        try:
                            res_kx, _ = q.scan_buffer(t, flags=agh.COUNT)
                        finally:
                            del os.environ["AGH_FS_FAST"]
                    finally:
                        q.set_costs(1, 1, 1)
                    assert (res_k.n_matched, [(s, e) for s, e, _ in ms_k]) == want_c, (costs, case["pattern"], case["opts"], ti)
                    assert res_kc.n_matched == res_kx.n_matched == want_c[0], (costs, case["pattern"], case["opts"], ti)
        q.close()


@pytest.mark.parametrize("delim", [b"s\n", b"\n", b"q"])
def test_table_engine_hands_long_records_over(agh, monkeypatch, delim):
    """Round 6: the fast table kernels walk past a chunk's end only until few lanes still have an open record and
    hand those to k_table_cont (AGH_TF_CONT lanes; 0: the old walk), and count-only scans of patterns without ';'
    count a flagged piece with one record end in the fast kernel (AGH_TF_DIRECT).  Records of ~1.7 KB with lengths
    all over the place ('s' + newline as the delimiter of the corpus), lines, and records of ~100 KB ('q'): every
    threshold, one and two streams per lane, the three chunk sizes, with costs, ';' and ',' -- always the oracle's
    count and record list."""
    text, _ = O.corpus(2048, seed=66, variants=(b"approxQmatch", b"approx--match", b"approximatematch", b"apprXmatZch",
                                                  b"match approx", b"aproxQmatch", b"approxmatch"), plant_period=7)
    tb_text = text.tobytes()                                        # 8 MiB
    cases = [(b"approx#match", 0), (b"approx#match", 1), (b"appr#mat#ch", 2), (b"match,approx", 1), (b"approx;match", 1)]
    for pat, k in cases:
        tb = agh.compile_pattern(pat, delim=delim)
        ot = O.tables_from_golden({"Mask": list(tb.Mask), "Init0": tb.Init0, "Init1": tb.Init1, "NO_ERR_MASK": tb.NO_ERR_MASK,
                                   "endposition": tb.endposition, "D_endpos": tb.D_endpos, "wildmask": tb.wildmask,
                                   "AND": tb.AND}, tb.M, dlen=len(delim))
        want = O.asearch_tables(ot, k, tb_text, delim=delim, cap=400000)
        assert want[0] > 20
        want_costs = O.asearch_tables_costs(ot, k, (2, 1, 1), tb_text, delim=delim, cap=400000) if k and not tb.AND else None
        with agh.Query.pattern(pat, k, delim=delim) as q:
            for env in ({"AGH_TF_CONT": "0"}, {"AGH_TF_CONT": "1"}, {"AGH_TF_CONT": "48"}, {"AGH_TF_CONT": "64"},
                        {"AGH_TF_CONT": "64", "AGH_TF_CHUNK": "1024"}, {"AGH_TF_CONT": "16", "AGH_TF_CHUNK": "4096"},
                        {"AGH_TF_CHUNK": "8192"}, {"AGH_TF_CHUNK": "16384", "AGH_TF_PACK2": "0"},
                        {"AGH_TF_CONT": "48", "AGH_TF_PACK2": "0"}, {"AGH_TF_CONT": "48", "AGH_TF_DIRECT": "0"},
                        {"AGH_TF_CONT": "0", "AGH_TF_DIRECT": "0"}):
                for key in ("AGH_TF_CONT", "AGH_TF_CHUNK", "AGH_TF_PACK2", "AGH_TF_DIRECT"):
                    monkeypatch.delenv(key, raising=False)
                for key, v in env.items():
                    monkeypatch.setenv(key, v)
                res, ms = q.scan_buffer(tb_text, cap=400000)
                assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (delim, pat, k, env)
                assert int(res.engine) == agh.ENGINE_FULLSCAN
                res_c, _ = q.scan_buffer(tb_text, flags=agh.COUNT)
                assert res_c.n_matched == want[0], (delim, pat, k, env, "count-only")
                if want_costs is not None and env.get("AGH_TF_CONT") in ("0", "48"):
                    q.set_costs(2, 1, 1)
                    try:
                        res_k, ms_k = q.scan_buffer(tb_text, cap=400000)
                        res_kc, _ = q.scan_buffer(tb_text, flags=agh.COUNT)
                    finally:
                        q.set_costs(1, 1, 1)
                    assert (res_k.n_matched, [(s, e) for s, e, _ in ms_k]) == want_costs, (delim, pat, k, env, "costs")
                    assert res_kc.n_matched == want_costs[0], (delim, pat, k, env, "costs, count-only")
    for key in ("AGH_TF_CONT", "AGH_TF_CHUNK", "AGH_TF_PACK2", "AGH_TF_DIRECT"):
        monkeypatch.delenv(key, raising=False)


def test_table_engine_multi_byte_delimiters_above_one_segment(agh, monkeypatch):
    """The table engine's fast form under a delimiter of several bytes on a text of several kernel segments
    (AGH_SEG_MAX_MB=1): the delimiter-end bitmap of the whole text, every segment on its part of it (16-byte loads at
    8-byte-aligned word offsets), segments whose cut is not aligned on a copy with a bitmap of their own; counts,
    records and record numbers against asearch.c on the same tables, fast forms and exact kernel."""
    monkeypatch.setenv("AGH_SEG_MAX_MB", "1")
    rng = random.Random(77)
    base, _ = O.corpus(1100, seed=21, variants=O.VARIANTS_C2[:5] + (b"approxXXmatch", b"aprox mat ch"), plant_period=9)
    for delim, width in ((b"e ", 0), (b"\n\n", 0), (b"ab", 63), (b"xyz", 64)):
        text = base.tobytes()
        if width:                               # records of `width` bytes: cuts fall on unaligned / aligned offsets
            body = text.replace(b"\n", b" ")
            text = delim.join(body[i:i + width - len(delim)] for i in range(0, 3400000, width - len(delim)))
        elif delim != b"e ":
            text = text.replace(b"\n", delim)
        for pat, k in ((b"approx#match", 1), (b"match,approx", 0), (b"approx;match", 2)):
            tb = agh.compile_pattern(pat, delim=delim)      # (== the reference's maskgen: tests/test_pattern_compiler.py)
            ot = O.tables_from_golden({"Mask": list(tb.Mask), "Init0": tb.Init0, "Init1": tb.Init1, "NO_ERR_MASK": tb.NO_ERR_MASK,
                                       "endposition": tb.endposition, "D_endpos": tb.D_endpos, "wildmask": tb.wildmask,
                                       "AND": tb.AND}, tb.M, dlen=len(delim))
            want = O.asearch_tables(ot, k, text, delim=delim, cap=600000)
            with agh.Query.pattern(pat, k, delim=delim) as q:
                res, ms = q.scan_buffer(text, cap=600000)
                res_c, _ = q.scan_buffer(text, flags=agh.COUNT)
                monkeypatch.setenv("AGH_FS_FAST", "0")
                res_x, _ = q.scan_buffer(text, flags=agh.COUNT)
                monkeypatch.delenv("AGH_FS_FAST")
            assert res.n_segments >= 3, (delim, res.n_segments)
            assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (delim, pat, k, res.copied_segments)
            assert res_c.n_matched == res_x.n_matched == want[0], (delim, pat, k)
            assert [i for _, _, i in ms] == sorted(i for _, _, i in ms)
    del rng


def test_table_engine_fuzz_delimiters_and_costs(agh):
    """Random `#` / `,` / `;` patterns under random delimiters (one byte, several bytes, sharing letters with the pattern,
    overlapping themselves), unit and edit costs, texts with delimiters across the 1 / 4 / 64 KiB borders of the fast
    forms: every form of the table engine (fast forms with one and two streams per lane, exact kernel; records, counts,
    numbered counts) against asearch.c / asearch1.c on the tables agh_compile_pattern makes (== maskgen's,
    tests/test_pattern_compiler.py).  AGH_FUZZ_SEED: another walk."""
    rng = random.Random(int(os.environ.get("AGH_FUZZ_SEED", "7")))
    done = 0
    for it in range(60):
        alpha = bytes(rng.sample(range(97, 123), rng.choice((3, 5, 8))))
        parts = [bytes(rng.choice(alpha) for _ in range(rng.randint(1, 4))) for _ in range(rng.randint(2, 3))]
        sep = rng.choice((b"#", b",", b";"))
        pat = sep.join(parts)
        k = rng.randint(0, 2)
        if rng.random() < 0.4:
            delim = b"\n"
        else:
            dl = rng.randint(2, 4)
            delim = bytes(rng.choice(alpha + b" .") for _ in range(dl))
            if rng.random() < 0.3:
                delim = delim[:1] * dl                         # overlaps itself
        try:
            tb = agh.compile_pattern(pat, delim=delim)
        except agh.AghError:
            continue                                            # (a pattern the reference turns down as well)
        if tb.M + len(delim) > 30:
            continue
        ot = O.tables_from_golden({"Mask": list(tb.Mask), "Init0": tb.Init0, "Init1": tb.Init1, "NO_ERR_MASK": tb.NO_ERR_MASK,
                                   "endposition": tb.endposition, "D_endpos": tb.D_endpos, "wildmask": tb.wildmask,
                                   "AND": tb.AND}, tb.M, dlen=len(delim))
        n = rng.choice((3000, 70000, 200000))
        body = bytearray(rng.choice(alpha + b"  ") for _ in range(n))
        for _ in range(n // 90):
            at = rng.randrange(n)
            body[at:at + len(delim)] = delim
        for _ in range(n // 400):                               # near matches
            at = rng.randrange(n)
            v = bytearray(b"".join(parts) if rng.random() < 0.5 else parts[0] + bytes(rng.choice(alpha) for _ in range(rng.randint(0, 5))) + parts[-1])
            if v and rng.random() < 0.5:
                v[rng.randrange(len(v))] = rng.choice(alpha)
            body[at:at + len(v)] = v
        for border in (1024, 4096, 65536, 131072):
            for shift in range(len(delim) + 1):
                at = border - shift
                if 0 <= at and at + len(delim) < n:
                    body[at:at + len(delim)] = delim
        text = bytes(body[:n]) + (delim if rng.random() < 0.5 else b"")
        costs = None if rng.random() < 0.6 or k == 0 else (rng.randint(1, 2), rng.randint(1, 2), rng.randint(1, 2))
        if costs:
            want = O.asearch_tables_costs(ot, k, costs, text, delim=delim, cap=400000)
        else:
            want = O.asearch_tables(ot, k, text, delim=delim, cap=400000)
        try:
            q = agh.Query.pattern(pat, k, delim=delim)
        except agh.AghError:
            continue                                            # (e.g. a pattern that matches the empty record with k errors)
        with q:
            if costs:
                q.set_costs(*costs)
            for env in ({}, {"AGH_TF_PACK2": "0"}, {"AGH_FS_FAST": "0"}, {"AGH_TF_CHUNK": "1024"}):
                os.environ.update(env)
                try:
                    res, ms = q.scan_buffer(text, cap=400000)
                    res_c, _ = q.scan_buffer(text, flags=agh.COUNT)
                    res_n, _ = q.scan_buffer(text, flags=agh.COUNT | agh.FORCE_NUMBERED)
                finally:
                    for key in env:
                        del os.environ[key]
                assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (env, pat, k, delim, costs, it)
                assert res_c.n_matched == res_n.n_matched == want[0], (env, pat, k, delim, costs, it)
        done += 1
    assert done >= 20


def test_piece_engine_for_short_patterns(agh):
    """Patterns the sample lemma cannot filter (m < 5k+6) run through the piece engine: k+1
    verbatim pieces found by the multi-pattern sweep, the pattern's automaton on the window.
    Every engine (pieces, full scan, lean, numbered) equals the oracle."""
    rng = random.Random(11)
    text, _ = O.corpus(256, seed=3, variants=O.VARIANTS_C2, plant_period=25, upper_permille=300)
    tb = bytearray(text.tobytes())
    words = [b"match", b"mat ch", b"aproxim", b"approxim", b"Approxim", b"appr", b"apr", b"mtch", b"matc",
             b"xmatch", b"approximate", b"aproximte"]
    pos = 0
    while True:                                         # sprinkle near-misses of the short words
        nl = tb.find(b"\n", pos)
        if nl < 0:
            break
        if rng.random() < 0.2 and nl - pos > 30:
            w = rng.choice(words)
            at = rng.randint(pos, nl - len(w))
            tb[at:at + len(w)] = w
        pos = nl + 1
    tb = bytes(tb)
    for pat, k, nocase in ((b"match", 1, False), (b"approxim", 1, False), (b"approxim", 2, True),
                           (b"appr", 0, False), (b"appr", 1, True), (b"approximate", 3, False),
                           (b"ab", 0, False), (b"abc", 1, False), (b"approximatematch", 4, True),
                           (b"approximatematch", 7, False)):
        with agh.Query(pat, k, nocase=nocase) as q:
            assert q.info()["filter_q"] == 0            # not sample-filterable
        res = _check(agh, pat, k, tb, nocase)
        assert res.engine in (agh.ENGINE_FILTER, agh.ENGINE_FULLSCAN)
    for t in (b"", b"match", b"mtch\n", b"x\nmatc", b"m\natch\n"):   # tiny texts, virtual head / tail
        _check(agh, b"match", 1, t)


def test_dense_hits_with_partial_last_strip(agh):
    """Two-letter alphabet: every position is a candidate, the piece engine checks hits inline
    (dense mode); the text length is not a multiple of 1 KiB, so the last strip takes the
    slice path with record numbers relative to its range (regression: they were absolute)."""
    rng = np.random.default_rng(5)
    for n in (600000, 262144 + 77, 1500):
        a = rng.integers(0, 2, n).astype(np.uint8) + ord("a")
        a[rng.integers(0, n, n // 80)] = 10
        text = a.tobytes()
        for pat, k in ((b"aab", 0), (b"bbabba", 1), (b"bbabbabbababaa", 2)):
            _check(agh, pat, k, text)


def test_inverse_is_the_complement_of_the_record_set(agh):
    """AGH_INVERT (-v, asearch.c:128): count = records - matched, list = the other records,
    for the filter, piece, full-scan and table engines, with and without a trailing newline."""
    text, _ = O.corpus(96, seed=21, variants=O.VARIANTS_C2, plant_period=7)
    base = text.tobytes()
    for tb in (base, base[:-1], b"", b"\n", b"\n\nabc\n\napproximatematch\n\n", b"approximatematch"):
        nl = [i for i, c in enumerate(tb) if c == 10]
        starts = [0] + [i + 1 for i in nl]
        ends = nl + [len(tb)]
        recs_all = [(s, e) for s, e in zip(starts, ends) if s < len(tb) or (s == len(tb) and False)]
        if tb.endswith(b"\n") or not tb:
            recs_all = list(zip(starts[:-1], ends[:-1]))
        for pat, k in ((O.PATTERN_C2, 2), (b"approxim", 1), (O.PATTERN_C2, 0)):
            hit = set(O.asearch(pat, k, tb, cap=200000)[1])
            want = [r for r in recs_all if r not in hit]
            with agh.Query(pat, k) as q:
                for extra in (0, agh.FORCE_FULLSCAN):
                    res, ms = q.scan_buffer(tb, flags=agh.INVERT | extra, cap=200000)
                    assert res.n_matched == len(want), (pat, k, len(tb), extra)
                    assert [(s, e) for s, e, _ in ms] == want
                    assert [i for _, _, i in ms] == [recs_all.index(r) for r in want[:50]] + [i for _, _, i in ms][50:]
                rc, _ = q.scan_buffer(tb, flags=agh.INVERT | agh.COUNT)
                assert rc.n_matched == len(want)


def test_general_queries_take_the_filter_path(agh):
    """Edit costs, <exact> segments and -w / -x guards keep the q-gram sample filter (samples
    from the literal core of the pattern) and verify with the general automaton: same records
    as the full scan and as the cost oracle, engine = filter."""
    text, _ = O.corpus(512, seed=8, variants=O.VARIANTS_C2, plant_period=9)
    tb = text.tobytes()
    for costs, k in (((2, 1, 1), 2), ((1, 2, 1), 2), ((1, 1, 2), 1)):
        want = O.asearch_costs(O.PATTERN_C2, k, costs, tb, cap=100000)
        with agh.Query(O.PATTERN_C2, k) as q:
            q.set_costs(*costs)
            res, ms = q.scan_buffer(tb, cap=100000)
            full, ms_f = q.scan_buffer(tb, cap=100000, flags=agh.FORCE_FULLSCAN)
            lean, _ = q.scan_buffer(tb, flags=agh.COUNT)
        assert res.engine == agh.ENGINE_FILTER and full.engine == agh.ENGINE_FULLSCAN
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want == (full.n_matched, [(s, e) for s, e, _ in ms_f])
        assert lean.n_matched == want[0]
    words = b"approximatematch\nxapproximatematch\napproximatematch x\nan aproximatematch!\n" * 50
    for case in _golden("pattern_language.json"):
        if case["pattern"] != "approximatematch":
            continue
        t = case["tables"]
        M = t["D_endpos"].bit_length()
        q = agh.Query.from_maskgen(t["Mask"], t["Init0"], t["Init1"], t["NO_ERR_MASK"], t["endposition"],
                                   t["D_endpos"], M, b"\n", case["k"], t["AND"])
        assert q.info()["filter_q"] >= 3                    # samples from the 16 literal positions
        for body in (tb + words, words):
            want = O.asearch_tables(O.tables_from_golden(t, M), case["k"], body, cap=100000)
            res, ms = q.scan_buffer(body, cap=100000)
            lean, _ = q.scan_buffer(body, flags=agh.COUNT)
            assert res.engine == agh.ENGINE_FILTER
            assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want and lean.n_matched == want[0]
        q.close()


def test_class_patterns_keep_the_sample_filter(agh):
    """A [class] inside the pattern no longer sends the query to the full scan: the q-grams of
    small classes are enumerated into the sample table (choose_filter / for_each_gram).  The
    reference's own maskgen tables in, filter engine == full scan == oracle out."""
    variants = (b"apprxximatematch", b"approximatematch", b"approxammatematch", b"approxmmatemmatch",
                b"aprroximatematch", b"approxzmatematch", b"apprximatematch")
    text = O.corpus(2048, seed=21, variants=variants, plant_period=9)[0].tobytes()
    seen = 0
    for case in _golden("pattern_language.json"):
        if case["pattern"] not in ("appr[ox]ximatematch", "approx[a-m]matematch"):
            continue
        t = case["tables"]
        M = t["D_endpos"].bit_length()
        q = agh.Query.from_maskgen(t["Mask"], t["Init0"], t["Init1"], t["NO_ERR_MASK"], t["endposition"],
                                   t["D_endpos"], M, b"\n", case["k"], t["AND"])
        info = q.info()
        assert info["filter_h"] > 0 and info["filter_q"] >= 3, (case["pattern"], info)
        assert (info["m"] - case["k"] - info["filter_q"] + 1) // info["filter_h"] >= case["k"] + 1
        want = O.asearch_tables(O.tables_from_golden(t, M), case["k"], text, cap=400000)
        res, ms = q.scan_buffer(text, cap=400000)
        full, ms_f = q.scan_buffer(text, cap=400000, flags=agh.FORCE_FULLSCAN)
        lean, _ = q.scan_buffer(text, flags=agh.COUNT)
        assert res.engine == agh.ENGINE_FILTER and full.engine == agh.ENGINE_FULLSCAN
        assert want[0] > 1000
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want == (full.n_matched, [(s, e) for s, e, _ in ms_f])
        assert lean.n_matched == want[0]
        q.close()
        seen += 1
    assert seen == 2


def test_streaming_count_scans_carry_records_across_segments(agh, tmp_path):
    """Count-only file scans stream through one device segment (agh_scan_fd -> stream_scan): the
    segment is cut after the last delimiter that has arrived and the unfinished record opens the
    next one (fill_buf's residue carry, sgrep.c:465-471).  An 80 MiB input whose 32 MiB read chunks
    end in the middle of records, from a file, from a pipe, with and without a trailing newline,
    and -l stopping early -- always the count of the whole-file scan."""
    import threading
    body, planted = O.corpus(20480, seed=77, variants=O.VARIANTS_C2, plant_period=40)      # 80 MiB
    head = b"x" * 1234 + b" approximatematch at the very top\n"
    for tail in (b"", b"and a last record without a newline: aproximatematch"):
        data = head + body.tobytes() + tail
        f = tmp_path / "stream.txt"
        f.write_bytes(data)
        with agh.Query(O.PATTERN_C2, 2) as q:
            fd = os.open(str(f), os.O_RDONLY)
            try:
                whole, ms = q.scan_fd(fd, cap=200000)                      # staged as one piece
                os.lseek(fd, 0, os.SEEK_SET)
                os.environ["AGH_STREAM_SEG_MB"] = "1"                      # a segment per read chunk
                try:
                    st, _ = q.scan_fd(fd, flags=agh.COUNT)
                    os.lseek(fd, 0, os.SEEK_SET)
                    st_l, _ = q.scan_fd(fd, flags=agh.FILENAMEONLY)
                finally:
                    del os.environ["AGH_STREAM_SEG_MB"]
                os.lseek(fd, 0, os.SEEK_SET)
                one, _ = q.scan_fd(fd, flags=agh.COUNT)                    # default 1 GiB segment
            finally:
                os.close(fd)
            want = int(sum(planted)) + 1 + (1 if tail else 0)
            assert whole.n_matched == len(ms) == want
            assert st.n_matched == want and st.n_segments >= 3 and st.n_bytes == len(data)
            assert one.n_matched == want
            assert st_l.n_matched >= 1 and st_l.n_bytes < len(data)         # stopped at the first hit
            # the same bytes through a pipe (no size known in advance, short reads)
            r, w = os.pipe()

            def feed():
                mv = memoryview(data)
                for i in range(0, len(mv), 1 << 20):
                    os.write(w, mv[i:i + (1 << 20)])
                os.close(w)
            th = threading.Thread(target=feed)
            th.start()
            os.environ["AGH_STREAM_SEG_MB"] = "1"
            try:
                pp, _ = q.scan_fd(r, flags=agh.COUNT)
            finally:
                del os.environ["AGH_STREAM_SEG_MB"]
                th.join()
                os.close(r)
            assert pp.n_matched == want and pp.n_bytes == len(data)


def test_nocase_with_a_letter_as_delimiter(agh):
    """-i -d q: maskgen.c:259-266 aliases the upper-case rows of Mask[] for the delimiter position
    too, so 'Q' ends a record as well.  One-byte delimiters that fold go through the delimiter
    bitmap like the multi-byte ones; all engines against the oracle (== the reference, checked on
    CPU: 57 / 83 records at k = 1 / 2 on this text)."""
    t, _ = O.corpus(16, seed=9, variants=O.VARIANTS_C2, plant_period=7, upper_permille=300)
    text = t.tobytes().replace(b"\n", b"q")
    assert text.count(b"Q") > 100
    for k, ref_count in ((1, 57), (2, 83), (0, None)):
        want = O.asearch(O.PATTERN_C2, k, text, delim=b"q", nocase=True, cap=100000)
        if ref_count is not None:
            assert want[0] == ref_count
        for flags in (0, agh.FORCE_FULLSCAN):
            res, recs, _ = _gpu(agh, O.PATTERN_C2, k, text, nocase=True, flags=flags, delim=b"q")
            assert (res.n_matched, recs) == want, (k, flags)
        with agh.Query(O.PATTERN_C2, k, nocase=True, delim=b"q") as q:
            assert q.scan_buffer(text, flags=agh.COUNT)[0].n_matched == want[0]
            assert q.scan_buffer(text, flags=agh.COUNT | agh.FORCE_FULLSCAN)[0].n_matched == want[0]
    rng = random.Random(99)
    for it in range(20):                                # small random cases, delimiter 'x' / 'X'
        pat, k, text = _rand_case(rng, 6, max_m=20)
        if b"x" in pat or b"X" in pat:
            continue
        text = bytes(c - 32 if (97 <= c <= 122 and rng.random() < 0.3) else c for c in text)
        text = text.replace(b"\n", b"x" if rng.random() < 0.5 else b"X")
        want = O.asearch(pat, k, text, delim=b"x", nocase=True, cap=100000)
        res, recs, _ = _gpu(agh, pat, k, text, nocase=True, delim=b"x")
        assert (res.n_matched, recs) == want, (pat, k, text)


@pytest.mark.parametrize("delim,nocase", [(b"\n\n", False), (b"From ", True), (b"q", True)])
def test_bitmap_delimiters_above_one_segment(agh, monkeypatch, delim, nocase):
    """Delimiters that come from the delimiter bitmap (several bytes, or a folded letter) on an
    input above the segment limit: the bitmap is built once for the whole text, the cuts sit at
    64-byte aligned delimiter ends, every segment works on its part of the bitmap.  Segmented ==
    unsegmented == the oracle, with record lists (absolute positions and numbers)."""
    text, _ = O.corpus(1536, seed=44, variants=O.VARIANTS_C2, plant_period=9,
                       upper_permille=300 if nocase else 0)                    # 6 MiB
    tb = text.tobytes().replace(b"\n", delim)
    for k in (0, 2):
        want = O.asearch(O.PATTERN_C2, k, tb, delim=delim, nocase=nocase, cap=400000)
        with agh.Query(O.PATTERN_C2, k, nocase=nocase, delim=delim) as q:
            one, ms1 = q.scan_buffer(tb, cap=400000)
            monkeypatch.setenv("AGH_SEG_MAX_MB", "1")
            try:
                seg, ms = q.scan_buffer(tb, cap=400000)
                seg_c, _ = q.scan_buffer(tb, flags=agh.COUNT)
                seg_f, _ = q.scan_buffer(tb, flags=agh.FORCE_FULLSCAN | agh.COUNT)
            finally:
                monkeypatch.delenv("AGH_SEG_MAX_MB")
        assert (one.n_matched, [(s, e) for s, e, _ in ms1]) == want
        assert seg.n_segments >= 4
        assert (seg.n_matched, [(s, e) for s, e, _ in ms]) == want, (delim, k)
        assert [i for _, _, i in ms] == [i for _, _, i in ms1]
        assert seg_c.n_matched == want[0] == seg_f.n_matched and seg.n_records == one.n_records


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_record_aligned_shards_add_up(agh, tmp_path, nranks):
    """What `agrep-hip --gpus N` and a one-process-per-GPU job do with a file, on the one GPU there
    is: agh_shard_cuts_fd cuts it into N record-aligned byte ranges, agh_scan_fd_range scans each
    (count-only: streaming; with records: staged), counts add up and the record lists, shifted by
    the shard offsets / the records in front, concatenate to the whole-file answer."""
    body, _ = O.corpus(700, seed=90 + nranks, variants=O.VARIANTS_C2, plant_period=13)
    data = b"no newline in front\n" + body.tobytes()[:-7]           # ragged: no trailing newline
    f = tmp_path / "shard.txt"
    f.write_bytes(data)
    fd = os.open(str(f), os.O_RDONLY)
    try:
        cuts = agh.shard_cuts_fd(fd, nranks)
        assert cuts[0] == 0 and cuts[-1] == len(data) and cuts == sorted(cuts)
        for c in cuts[1:-1]:
            assert c == len(data) or data[c - 1:c] == b"\n"
        with agh.Query(O.PATTERN_C2, 2) as q:
            whole, ms = q.scan_fd(fd, cap=100000)
            total, recs, rec_off = 0, [], 0
            for r in range(nranks):
                cnt, _ = q.scan_fd_range(fd, cuts[r], cuts[r + 1], flags=agh.COUNT)
                part, pm = q.scan_fd_range(fd, cuts[r], cuts[r + 1], cap=100000)
                assert cnt.n_matched == part.n_matched == len(pm)
                total += cnt.n_matched
                recs += [(s + cuts[r], e + cuts[r], i + rec_off) for s, e, i in pm]
                rec_off += part.n_records
    finally:
        os.close(fd)
    assert total == whole.n_matched and recs == ms and rec_off == whole.n_records
    assert (whole.n_matched, [(s, e) for s, e, _ in ms]) == O.asearch(O.PATTERN_C2, 2, data, cap=100000)


@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_fused_count_on_candidate_dense_text(agh, k, monkeypatch):
    """Every sample is a candidate (the text is made of the pattern's own grams): the verifying
    waves of the fused kernel are the bottleneck, the ring runs full, the hash set overflows into
    the numbered re-run -- and the count still equals the numbered pipeline's and the oracle's."""
    import torch
    rng = np.random.default_rng(77 + k)
    pat = O.PATTERN_C2
    parts = []
    for _ in range(40000):
        a = int(rng.integers(0, len(pat) - 4))
        b = int(rng.integers(a + 3, len(pat) + 1))
        parts.append(pat[a:b])
        if rng.random() < 0.2:
            parts.append(b"\n")
        if rng.random() < 0.05:
            parts.append(pat)
    text = b"".join(parts) + b"\n"
    want = O.asearch(pat, k, text)[0]
    with agh.Query(pat, k) as q:
        r_f, _ = q.scan_buffer(text, flags=agh.COUNT)
        r_n, _ = q.scan_buffer(text, flags=agh.COUNT | agh.FORCE_NUMBERED)
    assert r_f.n_matched == r_n.n_matched == want
    # the same text many times over on the device (64 MiB): fused against two kernels
    reps = (64 << 20) // len(text)
    t = torch.frombuffer(bytearray(text * reps), dtype=torch.uint8).cuda()
    monkeypatch.setenv("AGH_FUSED_MIN_MB", "0")         # (whatever the suite runs under: fused at this size)
    with agh.Query(pat, k) as q:
        r1 = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT)
        os.environ["AGH_FUSED"] = "0"
        try:
            r2 = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT)
        finally:
            del os.environ["AGH_FUSED"]
    assert r1.n_matched == r2.n_matched == want * reps
    if k <= 2:                                  # (k = 3 at m = 16 is the piece engine)
        assert r1.n_candidates == r2.n_candidates
        assert r1.fused_segments == 1 and r2.fused_segments == 0
        # the default policy keeps texts of this size on the two-kernel form
        saved = os.environ.pop("AGH_FUSED_MIN_MB")
        try:
            with agh.Query(pat, k) as q:
                r3 = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT)
        finally:
            os.environ["AGH_FUSED_MIN_MB"] = saved
        assert r3.fused_segments == 0 and r3.n_matched == r1.n_matched


def test_h2_sample_shape_parity(agh, monkeypatch):
    """The H = 2 / q = 4 sample shape (4-byte samples at every even offset, overlapping; lossless iff
    floor((m-k-3)/2) >= 2k+1: agh_query.cpp choose_filter): every pipeline against the oracle, the
    planted occurrences at every alignment, across strip / range / text-end boundaries."""
    monkeypatch.setenv("AGH_SHAPE_H2", "1")
    monkeypatch.setenv("AGH_FUSED_MIN_MB", "0")
    with agh.Query(O.PATTERN_C2, 2) as q:
        assert (q.info()["filter_q"], q.info()["filter_h"]) == (4, 2)
    with agh.Query(b"abcdefghij", 1) as q:              # no h >= 4 shape exists for (10, 1)
        assert (q.info()["filter_q"], q.info()["filter_h"]) == (4, 2)
    with agh.Query(O.PATTERN_C2, 0) as q:               # q = 4 already: the larger stride stays
        assert (q.info()["filter_q"], q.info()["filter_h"]) == (4, 8)
    for nocase in (False, True):
        text, planted = O.corpus(512, seed=4321, variants=O.VARIANTS_C2, plant_period=40,
                                 upper_permille=500 if nocase else 0)
        res = _check(agh, O.PATTERN_C2, 2, text, nocase)
        assert res.engine == agh.ENGINE_FILTER
    rng = random.Random(22)
    for _ in range(40):
        pat, k, text = _rand_case(rng, 6, max_m=29)
        _check(agh, pat, k, text)
    # occurrences (0..2 edits) at every byte alignment around strip (1 KiB), supertile (4 KiB) and wave
    # range (256 KiB) boundaries and at the very end of the text
    pat = O.PATTERN_C2
    variants = [pat, pat[:5] + pat[6:], pat[:7] + b"X" + pat[7:], pat[:3] + b"Q" + pat[4:9] + pat[10:]]
    for boundary in (1024, 4096, 262144):
        for v in variants:
            t = bytearray(b"z" * (boundary + 3000))
            for i in range(50, len(t), 97):
                t[i] = 10
            for shift in range(0, 24):
                tt = bytearray(t)
                at = boundary - shift
                tt[at:at + len(v)] = v
                for p in range(at - 1, at + len(v) + 1):
                    if tt[p] == 10:
                        tt[p] = ord("z")
                tb = bytes(tt)
                want = O.asearch(pat, 2, tb, cap=1000)
                assert want[0] == 1
                with agh.Query(pat, 2) as q:
                    r1, ms = q.scan_buffer(tb, cap=1000)
                    r2, _ = q.scan_buffer(tb, flags=agh.COUNT)
                assert (r1.n_matched, [(s, e) for s, e, _ in ms]) == want, (boundary, shift, v)
                assert r2.n_matched == 1, (boundary, shift, v)
    for tail in range(0, 40):
        tb = b"y" * 3000 + b"\n" + b"w" * tail + pat
        for cut in range(0, 3):
            tt = tb[:len(tb) - cut] if cut else tb
            want = O.asearch(pat, 2, tt, cap=100)
            with agh.Query(pat, 2) as q:
                r1, ms = q.scan_buffer(tt, cap=100)
                r2, _ = q.scan_buffer(tt, flags=agh.COUNT)
            assert (r1.n_matched, [(s, e) for s, e, _ in ms]) == want, (tail, cut)
            assert r2.n_matched == want[0], (tail, cut)


def test_fullscan_fast_form_equals_exact_kernel(agh, monkeypatch):
    """Full scans of unit-cost queries run as k_fullscan_fast (two text streams per lane for m <= 16)
    + k_fullscan_replay; AGH_FS_FAST=0 is the one-kernel exact form.  Same records from both, also
    where the replay lists overflow (a match in every record) and the scan falls back."""
    text, _ = O.corpus(700, seed=99, variants=O.VARIANTS_C2, plant_period=30)
    tb = text.tobytes()
    rng = random.Random(5)
    long_pat = bytes(rng.choice(b"abcdefghij") for _ in range(40))
    cases = [(O.PATTERN_C2, 2), (O.PATTERN_C2, 0), (b"approxim", 2), (b"match", 1), (b"approximate", 3),
             (b"approximatematchapproxim", 3), (long_pat, 2), (b"ab", 1), (b"e", 0)]
    dense = b"xxapproximatematchxx\n" * 30000 + b"approximatematch"          # every record matches
    for pat, k in cases:
        for t in (tb, tb[:70000] + b"approximatemat", dense):
            if len(pat) <= 29:
                want = O.asearch(pat, k, t, cap=400000)
            else:
                want = O.wm_count(pat, k, t, word_bits=64, cap=400000)
            for fast in ("1", "0"):
                monkeypatch.setenv("AGH_FS_FAST", fast)
                with agh.Query(pat, k) as q:
                    r1, ms = q.scan_buffer(t, flags=agh.FORCE_FULLSCAN, cap=400000)
                    r2, _ = q.scan_buffer(t, flags=agh.FORCE_FULLSCAN | agh.COUNT)
                    r3, _ = q.scan_buffer(t, flags=agh.FORCE_FULLSCAN | agh.COUNT | agh.FORCE_NUMBERED)
                assert r1.engine == agh.ENGINE_FULLSCAN
                assert (r1.n_matched, [(s, e) for s, e, _ in ms]) == want, (pat, k, fast, len(t))
                assert r2.n_matched == r3.n_matched == want[0], (pat, k, fast, len(t))


@pytest.mark.parametrize("streams", ["0", "1", "2"])
def test_fullscan_fast_streams_per_lane(agh, monkeypatch, streams):
    """Round 6: k_fullscan_fast keeps THREE text streams per lane in the 10-bit fields of a 32-bit word for m <= 10
    (two 16-bit halves for m <= 16 since round 3).  Every stream count (AGH_FS_STREAMS caps it; 0: by the pattern's
    length) against asearch.c's restatement: patterns of 2..11 bytes, every k below the length
    up to 4 (and 5..7 on m = 8), texts whose tiles end inside occurrences, a text of one byte more than a group of
    tiles, the record list and both count-only forms."""
    monkeypatch.setenv("AGH_FS_STREAMS", streams)
    rng = random.Random(77)
    base, _ = O.corpus(400, seed=77, variants=(b"matching", b"matchng", b"maXching", b"mmatching", b"approxima", b"aproxima",
                                                  b"wordlength", b"wrdlength"), plant_period=9)
    tb = base.tobytes()
    tile = 64 * 1024
    texts = [tb, tb[:4 * tile + 1], tb[:3 * tile - 3] + b"matchin", tb[:tile] + b"\n" * 70000 + tb[:9000], b"matching", b"x"]
    pats = [b"matching", b"approxima", b"wordlength", b"match", b"at", b"the", b"tching", b"lengthwords", b"e tao ns"]
    n = 0
    for pat in pats:
        for k in (1, 2, 3, 4, 5, 7):
            if k >= len(pat) or (k > 4 and len(pat) != 8):
                continue
            for ti, t in enumerate(texts):
                if k > 2 and ti not in (0, 2):
                    continue
                want = O.asearch(pat, k, t, cap=600000)
                with agh.Query(pat, k) as q:
                    r1, ms = q.scan_buffer(t, flags=agh.FORCE_FULLSCAN, cap=600000)
                    r2, _ = q.scan_buffer(t, flags=agh.FORCE_FULLSCAN | agh.COUNT)
                assert r1.engine == agh.ENGINE_FULLSCAN
                assert (r1.n_matched, [(s_, e_) for s_, e_, _ in ms]) == want, (pat, k, streams, ti)
                assert r2.n_matched == want[0], (pat, k, streams, ti)
                n += 1
    assert n > 60


def test_table_engine_with_edit_costs(agh):
    """'#' / ';' / ',' together with -I -S -D: asearch1.c:88-97 runs on the same Init1 / endposition
    tables; the table engine's feed_costs against the oracle's restatement of asearch1.c on the
    reference's own tables (tests/golden/pattern_language.json)."""
    rng = np.random.default_rng(11)
    words = [b"car", b"cars", b"red", b"fast", b"scar", b"cat", b"a", b"ca r", b"cr", b"caar", b"rad", b" ", b"\n", b"\n"]
    text = b"".join(words[i] for i in rng.integers(0, len(words), 40000))
    text += O.corpus(96, seed=31, variants=O.VARIANTS_C2[:5] + (b"approxXXmatch", b"aprxmatch", b"matematch approx"),
                     plant_period=12)[0].tobytes()
    n = 0
    for case in _golden("pattern_language.json"):
        if case["k"] < 1 or not any(c in case["pattern"] for c in "#;,"):
            continue
        tb = case["tables"]
        M = tb["D_endpos"].bit_length()
        ot = O.tables_from_golden(tb, M)
        for costs in ((2, 1, 1), (1, 2, 1), (1, 1, 2), (2, 2, 2), (3, 1, 2)):
            q = agh.Query.from_maskgen(tb["Mask"], tb["Init0"], tb["Init1"], tb["NO_ERR_MASK"],
                                       tb["endposition"], tb["D_endpos"], M, b"\n", case["k"], tb["AND"])
            q.set_costs(*costs)
            want = O.asearch_tables_costs(ot, case["k"], costs, text, cap=200000)
            res, ms = q.scan_buffer(text, cap=200000)
            res_c, _ = q.scan_buffer(text, flags=agh.COUNT)
            # (round 5: costs take the fast form -- branch-free kernel + exact replay -- as well; the one-kernel form beside it)
            os.environ["AGH_FS_FAST"] = "0"
            try:
                res_x, ms_x = q.scan_buffer(text, cap=200000)
                res_xc, _ = q.scan_buffer(text, flags=agh.COUNT)
            finally:
                del os.environ["AGH_FS_FAST"]
            q.close()
            assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (case["pattern"], case["k"], costs)
            assert (res_x.n_matched, [(s, e) for s, e, _ in ms_x]) == want, ("exact kernel", case["pattern"], case["k"], costs)
            assert res_c.n_matched == want[0] == res_xc.n_matched
            n += 1
    assert n >= 5


def test_segments_cut_at_unaligned_record_ends(agh, monkeypatch):
    """Inputs above the segment limit are cut where a record ends; where no record ends on a 16-byte
    boundary -- 64-byte records behind a 1-byte header -- the cut is any record end and the segment
    is scanned from an aligned copy (agh_result.copied_segments).  Counts, records and record numbers
    equal the oracle's; the same with a two-byte delimiter (per-segment delimiter bitmap)."""
    monkeypatch.setenv("AGH_SEG_MAX_MB", "1")
    rng = random.Random(3)
    recs = []
    for i in range(70000):
        r = bytearray(rng.choice(b"abcdefgh ") for _ in range(63))
        if i % 97 == 0:
            v = bytearray(b"approximatematch")
            if i % 2:
                v[5] = ord("Z")
            at = rng.randrange(0, 63 - len(v))
            r[at:at + len(v)] = v
        recs.append(bytes(r))
    text = b"H" + b"\n".join(recs) + b"\n"
    assert all((1 + 64 * (j + 1)) % 16 for j in range(10))
    want = O.asearch(b"approximatematch", 2, text, cap=10000)
    with agh.Query(b"approximatematch", 2) as q:
        r1, ms = q.scan_buffer(text, cap=10000)
        r2, _ = q.scan_buffer(text, flags=agh.COUNT)
        r3, _ = q.scan_buffer(text, flags=agh.COUNT | agh.FORCE_FULLSCAN)
    assert r1.n_segments >= 4 and r1.copied_segments >= 3, (r1.n_segments, r1.copied_segments)
    assert (r1.n_matched, [(s, e) for s, e, _ in ms]) == want
    assert [i for _, _, i in ms] == [s // 64 for s, _ in want[1]]
    assert r2.n_matched == r3.n_matched == want[0]
    # -f on the same file
    with agh.Query.multi([b"approximatematch", b"approZimatematch"]) as q:
        rm, _ = q.scan_buffer(text, flags=agh.COUNT)
    assert rm.n_matched == want[0]
    # two-byte delimiter: 62-byte records + "\r\n" behind a 1-byte header never end on a 64-byte boundary
    t2 = b"H" + b"\r\n".join(r[:62] for r in recs[:50000]) + b"\r\n"
    want2 = O.asearch(b"approximatematch", 2, t2, delim=b"\r\n", cap=10000)
    with agh.Query(b"approximatematch", 2, delim=b"\r\n") as q:
        r4, ms4 = q.scan_buffer(t2, cap=10000)
        r5, _ = q.scan_buffer(t2, flags=agh.COUNT)
    assert r4.copied_segments >= 1
    assert (r4.n_matched, [(s, e) for s, e, _ in ms4]) == want2 and r5.n_matched == want2[0]


def test_inverse_record_list_with_multi_byte_delimiters(agh):
    """-v with record output where the delimiter has several bytes (or is a folded letter): k_unmatched
    reads the delimiter-end bitmap.  The complement of the oracle's record set, record by record."""
    text, _ = O.corpus(64, seed=23, variants=O.VARIANTS_C2, plant_period=5)
    base = text.tobytes()
    for delim, tb in ((b"\r\n", base.replace(b"\n", b"\r\n")), (b"$$", base.replace(b"\n", b"$$")[:-2]),
                      (b"e ", base.replace(b"\n", b" "))):
        # all records: the text between (leftmost, non-overlapping) delimiter occurrences
        recs_all, at = [], 0
        while True:
            i = tb.find(delim, at)
            if i < 0:
                if at < len(tb):
                    recs_all.append((at, len(tb)))
                break
            recs_all.append((at, i))
            at = i + len(delim)
        for pat, k in ((O.PATTERN_C2, 2), (b"approxim", 1)):
            hit = set(O.asearch(pat, k, tb, delim=delim, cap=200000)[1])
            want = [r for r in recs_all if r not in hit]
            with agh.Query(pat, k, delim=delim) as q:
                res, ms = q.scan_buffer(tb, flags=agh.INVERT, cap=200000)
                rc, _ = q.scan_buffer(tb, flags=agh.INVERT | agh.COUNT)
            assert res.n_matched == rc.n_matched == len(want), (delim, pat, k)
            assert [(s, e) for s, e, _ in ms] == want, (delim, pat, k)
            assert [i for _, _, i in ms][:40] == [recs_all.index(r) for r in want[:40]]


def test_stress_parity_slice():
    """A seeded 60-second slice of scripts/stress_parity.py (random patterns / k / -i / delimiters / texts /
    -f sets / table-engine cases, every device engine against the oracle) inside the suite, so that the
    driver's own run executes it; longer runs: profiles/r04_stress_parity*.log."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "stress_parity.py")
    p = subprocess.run([sys.executable, script, "60", "20260926"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    tail = p.stdout.decode(errors="replace").strip().splitlines()[-5:]
    assert p.returncode == 0 and tail and tail[-1].endswith("failures 0"), tail
    assert int(tail[-1].split()[1]) > 300, tail


def test_long_records_do_not_rerun_the_segment(agh):
    """Count-only scans identify a matched record by the offset of its first byte, found by looking back at
    most 1 MiB from the match.  Records longer than that used to send the whole segment to the numbered
    pipeline (agh_result.lean_reruns); now the match is noted and k_resolve_giveups finds its record start
    after the scan: one 3 MiB record, several matches in one 5 MiB record (counted once), a long first
    record (no delimiter in front of it at all), every count-only engine."""
    import torch
    base, _ = O.corpus(4096, seed=11, variants=O.VARIANTS_C2, plant_period=50)       # 16 MiB
    x = lambda n: np.full(n, ord("x"), dtype=np.uint8)
    lit = lambda b: np.frombuffer(b, dtype=np.uint8)
    text = np.concatenate([x(2 << 20), lit(b" approximatematch "), x(1 << 20), lit(b"\n"),          # long FIRST record
                           base[:8 << 20],
                           x(3 << 20), lit(b" aproximatematch\n"),                                 # 3 MiB in front of a match
                           base[8 << 20:],
                           lit(b"approximatematch "), x(2 << 20), lit(b" approximatematch "), x(3 << 20),
                           lit(b" approxXmatematch\n")])                                            # three matches, one record
    want = O.asearch(O.PATTERN_C2, 2, text)[0]
    t = torch.from_numpy(text).cuda()
    for flags in (agh.COUNT, agh.COUNT | agh.FORCE_FULLSCAN):
        with agh.Query(O.PATTERN_C2, 2) as q:
            r = q.scan_device(t.data_ptr(), t.numel(), flags=flags)
            rn = q.scan_device(t.data_ptr(), t.numel(), flags=flags | agh.FORCE_NUMBERED)
        assert r.n_matched == rn.n_matched == want and r.lean_reruns == 0, flags
    os.environ["AGH_FUSED"] = "0"                       # the two-kernel count-only form
    try:
        with agh.Query(O.PATTERN_C2, 2) as q:
            r = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT)
    finally:
        del os.environ["AGH_FUSED"]
    assert r.n_matched == want and r.lean_reruns == 0
    with agh.Query.multi([b"approxim", b"matematch"], k=1) as q:         # the one-pass -f kernel
        r = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT)
        rn = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT | agh.FORCE_NUMBERED)
    assert r.n_matched == rn.n_matched and r.lean_reruns == 0 and r.fused_segments == 1
