"""-f multi-pattern scans (newmgrep.c semantics: a record matches iff it contains any pattern
verbatim) against the naive oracle, and -- for <= 16 patterns, where the reference is reliable
 -- against the reference CLI."""
import os
import random
import subprocess

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "agrep_amd", "agrep-hip")
REF = os.path.join(O.REF_DIR, "agrep")


@pytest.fixture(scope="module")
def agh():
    import agrep_amd
    assert agrep_amd.device_count() >= 1
    return agrep_amd


def _rand_patterns(rng, n, lo, hi, alphabet=b"abcdefghijklmnopqrstuvwxyz"):
    out = set()
    while len(out) < n:
        out.add(bytes(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))))
    return sorted(out)


def _plant(text, pats, rng, every):
    a = bytearray(text)
    pos = 0
    while True:
        nl = a.find(b"\n", pos)
        if nl < 0:
            break
        if rng.random() < 1.0 / every and nl - pos > 40:
            p = rng.choice(pats)
            at = rng.randint(pos, nl - len(p))
            a[at:at + len(p)] = p
        pos = nl + 1
    return bytes(a)


def _check(agh, pats, text, nocase=False):
    want = O.multi_exact_count(pats, text, nocase=nocase, cap=200000)
    with agh.Query.multi(pats, nocase=nocase) as q:
        res, ms = q.scan_buffer(text, cap=200000)
        res_c, _ = q.scan_buffer(text, flags=agh.COUNT)
    assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want
    assert res_c.n_matched == want[0]
    return res


@pytest.mark.parametrize("npat,lo,hi", [(1, 6, 6), (16, 4, 12), (50, 4, 12), (1024, 4, 12), (300, 1, 3),
                                        (5, 13, 40)])
def test_multi_pattern_counts_and_records(agh, npat, lo, hi):
    rng = random.Random(npat * 7 + lo)
    pats = _rand_patterns(rng, npat, lo, hi)
    base, _ = O.corpus(96, seed=npat, variants=(), plant_period=0)
    text = _plant(base.tobytes(), pats, rng, every=9)
    res = _check(agh, pats, text)
    assert res.n_matched > 0


def test_multi_pattern_nocase_and_edges(agh):
    rng = random.Random(3)
    pats = [b"Needle", b"haystack", b"XyZ", b"q"]
    base, _ = O.corpus(32, seed=1, variants=(), plant_period=0, upper_permille=400)
    text = _plant(base.tobytes(), [b"nEEdle", b"HAYSTACK", b"xyz"], rng, every=5)
    _check(agh, pats[:3], text, nocase=True)
    _check(agh, pats[:3], text, nocase=False)
    for t in (b"", b"\n", b"needle", b"x needle", b"needle\n" * 3000, b"nee\ndle\n", b"a" * 5000 + b"needle"):
        _check(agh, [b"needle", b"zzz"], t)
    # patterns across strip / range boundaries
    for boundary in (1024, 4096, 262144):
        t = bytearray(b"z" * (boundary + 2048))
        for i in range(70, len(t), 91):
            t[i] = 10
        for shift in (1, 3, 5, 11):
            at = boundary - shift
            t[at:at + 6] = b"needle"
            for p in range(at - 1, at + 8):
                if t[p] == 10:
                    t[p] = ord("z")
        _check(agh, [b"needle", b"absent"], bytes(t))


def test_multi_pattern_resident_scale(agh):
    """Config-5 shape: 1024 patterns of 4..12 bytes, 1 GiB resident; lean == numbered and a
    16 MiB slice equals the oracle."""
    import torch
    rng = random.Random(1024)
    pats = _rand_patterns(rng, 1024, 4, 12)
    pages = (1 << 30) // 4096
    t = torch.empty(pages * 4096, dtype=torch.uint8, device="cuda")
    agh.corpus_fill_device(t.data_ptr(), pages, seed=5, variants=tuple(pats[:7]), plant_period=300)
    with agh.Query.multi(pats) as q:
        a = q.scan_device(t.data_ptr(), t.numel())
        b = q.scan_device(t.data_ptr(), t.numel(), flags=agh.COUNT)
        sl = q.scan_device(t.data_ptr(), 16 << 20, flags=agh.COUNT)
    assert a.n_matched == b.n_matched > 0
    assert sl.n_matched == O.multi_exact_count(pats, t[:16 << 20].cpu().numpy())[0]


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/agrep not built")
def test_cli_pattern_file_matches_reference(agh, tmp_path):
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "agrep_amd", "host")])
    rng = random.Random(16)
    pats = _rand_patterns(rng, 12, 5, 10)
    base, _ = O.corpus(40, seed=77, variants=(), plant_period=0)
    text = _plant(base.tobytes(), pats, rng, every=7)
    f = tmp_path / "text.txt"
    f.write_bytes(text)
    pf = tmp_path / "pats.txt"
    pf.write_bytes(b"\n".join(pats) + b"\n")
    for args in (["-V0", "-c"], ["-V0"], ["-V0", "-l"], ["-c"], ["-V0", "-i", "-c"]):
        a = args + ["-f", str(pf), str(f)]
        r = subprocess.run([REF] + a, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        g = subprocess.run([CLI] + a, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert g.stdout == r.stdout, (args, g.stdout[:200], r.stdout[:200])
        assert g.returncode == r.returncode, args


# ---- -f with errors (agh_query_multi_approx): union of the single-pattern k-error predicate ----
def _mutate(p, k, rng, alphabet=b"abcdefghijklmnopqrstuvwxyz"):
    a = bytearray(p)
    for _ in range(k):
        op = rng.randint(0, 2)
        at = rng.randint(0, max(0, len(a) - 1))
        if op == 0 and len(a) > 1:
            del a[at]
        elif op == 1:
            a.insert(at, rng.choice(alphabet))
        else:
            a[at] = rng.choice(alphabet)
    return bytes(a)


def _approx_want(pats, k, text, nocase=False):
    recs = set()
    for p in pats:
        recs.update(O.asearch(p, k, text, nocase=nocase, cap=400000)[1])
    return sorted(recs)


def _check_approx(agh, pats, k, text, nocase=False):
    want = _approx_want(pats, k, text, nocase)
    with agh.Query.multi(pats, nocase=nocase, k=k) as q:
        res, ms = q.scan_buffer(text, cap=400000)
        res_c, _ = q.scan_buffer(text, flags=agh.COUNT)
        res_n, _ = q.scan_buffer(text, flags=agh.COUNT | agh.FORCE_NUMBERED)
    assert [(s, e) for s, e, _ in ms] == want
    assert res.n_matched == res_c.n_matched == res_n.n_matched == len(want)
    return res


@pytest.mark.parametrize("npat,lo,hi,k", [(1, 8, 8, 1), (16, 6, 12, 1), (64, 8, 12, 2), (200, 4, 12, 1),
                                          (12, 10, 12, 3), (40, 2, 3, 1)])
def test_multi_pattern_with_errors(agh, npat, lo, hi, k):
    rng = random.Random(npat * 11 + k)
    pats = _rand_patterns(rng, npat, lo, hi)
    base, _ = O.corpus(48, seed=npat + k, variants=(), plant_period=0)
    planted = [_mutate(rng.choice(pats), rng.randint(0, k + 1), rng) for _ in range(64)]
    text = _plant(base.tobytes(), planted, rng, every=6)
    res = _check_approx(agh, pats, k, text)
    assert res.n_matched > 0


def test_multi_pattern_with_errors_edges(agh):
    rng = random.Random(5)
    pats = [b"needle", b"haystack", b"Thread"]
    for t in (b"", b"\n", b"nedle", b"x neeedle", b"needl", b"eedle\nhaystac\n\nthreat\n", b"need\nle\n",
              b"a" * 5000 + b"hay5tack", b"needl\n" * 3000):
        _check_approx(agh, pats, 1, t)
        _check_approx(agh, pats, 2, t, nocase=True)
    for boundary in (1024, 4096, 262144):           # occurrences across strip / range boundaries
        t = bytearray(b"z" * (boundary + 2048))
        for i in range(70, len(t), 91):
            t[i] = 10
        for shift in (1, 3, 5, 11):
            at = boundary - shift
            t[at:at + 5] = b"nedle"
            for p in range(at - 2, at + 8):
                if t[p] == 10:
                    t[p] = ord("z")
        _check_approx(agh, [b"needle", b"absent"], 1, bytes(t))
    with pytest.raises(agh.AghError):
        agh.Query.multi([b"ab", b"needle"], k=2)        # length must exceed k
    with pytest.raises(agh.AghError):
        agh.Query.multi([b"nee\ndle"], k=1)


@pytest.mark.parametrize("alphabet,lo,hi,nocase", [(b"ab", 2, 14, False), (b"abc", 3, 9, False), (b"aB", 4, 16, True),
                                                  (b"ab", 15, 20, False)])
def test_multi_pattern_one_error_small_alphabet(agh, alphabet, lo, hi, nocase):
    """k = 1: the side next to a verbatim piece is checked in two 64-bit words (agh_multi.hip
    side_within_one_edit) when it has <= 7 bytes, by the automaton otherwise.  Texts over the patterns'
    own tiny alphabet with short records: every position is a near miss, delimiters land on the
    replaced / extra byte, occurrences touch both ends of the text."""
    rng = random.Random(len(alphabet) * 100 + lo)
    pats = _rand_patterns(rng, 12, lo, hi, alphabet=alphabet)
    talpha = (alphabet + alphabet.swapcase() if nocase else alphabet) + b"\n"
    for n in (1, 7, 300, 70000):
        text = bytes(rng.choice(talpha) for _ in range(n))
        _check_approx(agh, pats, 1, text, nocase=nocase)
    text = bytes(rng.choice(alphabet + b"z") if i % 23 else 10 for i in range(200000))
    _check_approx(agh, pats, 1, text, nocase=nocase)


@pytest.mark.parametrize("npat,lo,hi,stride,nocase", [(100, 5, 9, 2, False), (500, 7, 12, 4, False),
                                                      (1024, 8, 12, 4, False), (64, 5, 6, 2, True),
                                                      (200, 7, 30, 4, True), (3, 25, 38, 4, False)])
def test_multi_pattern_strided_probing(agh, npat, lo, hi, stride, nocase):
    """Sets whose shortest entry has >= 5 (>= 7) bytes are probed at every 2nd (4th) text position
    only, with the 4-grams of entry offsets 0..S-1 in the table (fill_multi_tables): an entry that
    occurs verbatim contains a 4-gram at a position divisible by S.  Same answers as the oracle,
    occurrences planted at every alignment."""
    rng = random.Random(npat * 13 + lo)
    pats = _rand_patterns(rng, npat, lo, hi)
    with agh.Query.multi(pats, nocase=nocase) as q:
        assert q.info()["filter_h"] == stride
    base, _ = O.corpus(96, seed=npat + 1, variants=(), plant_period=0)
    text = _plant(base.tobytes(), pats, rng, every=5)
    if nocase:
        text = bytes(c - 32 if (97 <= c <= 122 and rng.random() < 0.4) else c for c in text)
    res = _check(agh, pats, text, nocase=nocase)
    assert res.n_matched > 50
    # every alignment of one pattern, at the very start and the very end of the text too
    p0 = pats[0]
    small = b"".join(b"x" * i + p0 + b"\n" for i in range(9))
    for t in (p0, p0 + b"\n", b"\n" + p0, small, small[:-1], b"y" * 3 + small):
        _check(agh, pats, t, nocase=nocase)


@pytest.mark.parametrize("npat,lo,hi,k,stride", [(30, 14, 24, 1, 4), (30, 10, 13, 1, 2), (20, 21, 29, 2, 4),
                                                  (1, 16, 16, 1, 4)])
def test_multi_pattern_with_errors_strided(agh, npat, lo, hi, k, stride):
    """-f with errors over long patterns: the k+1 pieces have >= 5 / >= 7 bytes, so the piece sweep
    probes every 2nd / 4th position; the pattern's automaton runs from the piece's start."""
    rng = random.Random(npat * 17 + k)
    pats = _rand_patterns(rng, npat, lo, hi)
    with agh.Query.multi(pats, k=k) as q:
        assert q.info()["filter_h"] == stride
    base, _ = O.corpus(48, seed=npat + k + 3, variants=(), plant_period=0)
    planted = [_mutate(rng.choice(pats), rng.randint(0, k + 1), rng) for _ in range(64)]
    text = _plant(base.tobytes(), planted, rng, every=6)
    res = _check_approx(agh, pats, k, text)
    assert res.n_matched > 0


@pytest.mark.parametrize("delim", [b"\r\n", b"$$", b"; ", b"\n\n"])
def test_multi_pattern_with_multi_byte_delimiters(agh, delim):
    """-f (exact and with errors) where records are separated by several bytes: record ends come from
    the delimiter bitmap, as in the single-pattern engines (preproce.c:181-224, delim.c:49-96 for
    mgrep's record bounds).  Records, record numbers and counts against the oracle."""
    rng = random.Random(len(delim) * 31 + delim[0])
    pats = _rand_patterns(rng, 40, 5, 11)
    base, _ = O.corpus(40, seed=9, variants=(), plant_period=0)
    planted = [_mutate(rng.choice(pats), rng.randint(0, 2), rng) for _ in range(64)]
    text = _plant(base.tobytes(), planted, rng, every=5).replace(b"\n", delim)
    for t in (text, text[:-len(delim)], text[:100000] + delim + delim + text[100000:200000]):
        want = O.multi_exact_count(pats, t, delim=delim, cap=200000)
        with agh.Query.multi(pats, delim=delim) as q:
            res, ms = q.scan_buffer(t, cap=200000)
            res_c, _ = q.scan_buffer(t, flags=agh.COUNT)
        assert (res.n_matched, [(s, e) for s, e, _ in ms]) == want, (delim, len(t))
        assert res_c.n_matched == want[0]
        recs = set()
        for p in pats:
            recs.update(O.asearch(p, 1, t, delim=delim, cap=400000)[1])
        want1 = sorted(recs)
        with agh.Query.multi(pats, delim=delim, k=1) as q:
            res, ms = q.scan_buffer(t, cap=400000)
            res_c, _ = q.scan_buffer(t, flags=agh.COUNT)
            res_n, _ = q.scan_buffer(t, flags=agh.COUNT | agh.FORCE_NUMBERED)
        assert [(s, e) for s, e, _ in ms] == want1, (delim, len(t))
        assert res.n_matched == res_c.n_matched == res_n.n_matched == len(want1)


def test_multi_pattern_nocase_letter_delimiter(agh):
    """-i -d q -f: 'Q' ends a record as well (maskgen.c:259-266 aliases the delimiter's rows too)."""
    rng = random.Random(17)
    pats = [b"needle", b"haystack", b"stall"]
    base, _ = O.corpus(24, seed=3, variants=(b"Needle", b"HAYSTACK", b"needle"), plant_period=6, upper_permille=200)
    t = base.tobytes().replace(b"\n", b"q")
    recs = set()
    for p in pats:
        recs.update(O.asearch(p, 0, t, delim=b"q", nocase=True, cap=100000)[1])
    want = sorted(recs)
    with agh.Query.multi(pats, nocase=True, delim=b"q") as q:
        res, ms = q.scan_buffer(t, cap=100000)
        res_c, _ = q.scan_buffer(t, flags=agh.COUNT)
    assert [(s, e) for s, e, _ in ms] == want and res.n_matched == res_c.n_matched == len(want)


# ---- the one-pass count-only scan (agh_mscan.hip): sets it takes, every boundary it has ----------
def _one_pass_count(agh, pats, k, text, nocase=False, expect=True):
    """COUNT scan (the one-pass kernel where the set qualifies) == numbered pipeline -- for the dense one-error sets
    both on the tile kernel's numbered form (marks by record number) and on round 5's k_dense_multi
    (AGH_MTILE_NUMBERED=0: another kernel, another verifier)."""
    with agh.Query.multi(pats, nocase=nocase, k=k) as q:
        c, _ = q.scan_buffer(text, flags=agh.COUNT)
        n, _ = q.scan_buffer(text, flags=agh.COUNT | agh.FORCE_NUMBERED)
        saved = os.environ.get("AGH_MTILE_NUMBERED")
        os.environ["AGH_MTILE_NUMBERED"] = "0"
        try:
            n0, _ = q.scan_buffer(text, flags=agh.COUNT | agh.FORCE_NUMBERED)
        finally:
            if saved is None:
                del os.environ["AGH_MTILE_NUMBERED"]
            else:
                os.environ["AGH_MTILE_NUMBERED"] = saved
    assert c.n_matched == n.n_matched == n0.n_matched, (len(text), k, nocase, c.n_matched, n.n_matched, n0.n_matched)
    if len(text):
        assert (c.fused_segments == 1) == expect, "one-pass kernel %s" % ("did not run" if expect else "ran")
        assert c.lean_reruns == 0
    return c.n_matched


@pytest.mark.parametrize("rb", ["13", "12"])
def test_one_pass_scan_takes_the_sets_it_should(agh, rb, monkeypatch):
    monkeypatch.setenv("AGH_MSCAN_RB", rb)
    rng = random.Random(41 + int(rb))
    base, _ = O.corpus(160, seed=7, variants=(), plant_period=0)            # 640 KiB: three wave ranges
    p812 = _rand_patterns(rng, 300, 8, 12)
    planted = [_mutate(rng.choice(p812), rng.randint(0, 2), rng) for _ in range(200)]
    text = _plant(base.tobytes(), planted, rng, every=4)
    want1 = len(_approx_want(p812, 1, text))
    assert _one_pass_count(agh, p812, 1, text) == want1 > 100
    # (exact entries of >= 7 bytes are probed at every 4th position by the two-kernel form: faster there)
    assert _one_pass_count(agh, p812, 0, text, expect=False) == O.multi_exact_count(p812, text)[0] > 10
    p512 = _rand_patterns(rng, 300, 5, 12)
    text5 = _plant(base.tobytes(), p512, rng, every=5)
    assert _one_pass_count(agh, p512, 0, text5) == O.multi_exact_count(p512, text5)[0] > 100
    p414 = _rand_patterns(rng, 200, 4, 15)
    text0 = _plant(base.tobytes(), p414, rng, every=5)
    assert _one_pass_count(agh, p414, 0, text0) == O.multi_exact_count(p414, text0)[0] > 100
    # -i: mixed-case text, patterns with capitals
    up = [bytes(c - 32 if rng.random() < 0.3 else c for c in p) for p in p812[:100]]
    mixed = bytes(c - 32 if (97 <= c <= 122 and rng.random() < 0.3) else c for c in text)
    assert _one_pass_count(agh, up, 1, mixed, nocase=True) == len(_approx_want(up, 1, mixed, nocase=True)) > 50
    # sets it leaves to the two-kernel form: k = 2, patterns above 14 bytes with an error, above 15 without,
    # entries below 4 bytes
    _one_pass_count(agh, _rand_patterns(rng, 40, 9, 12), 2, text, expect=False)
    _one_pass_count(agh, _rand_patterns(rng, 40, 10, 16), 1, text, expect=False)
    _one_pass_count(agh, _rand_patterns(rng, 40, 6, 20), 0, text, expect=False)
    _one_pass_count(agh, _rand_patterns(rng, 40, 3, 9), 0, text, expect=False)
    monkeypatch.setenv("AGH_MSCAN", "0")
    _one_pass_count(agh, p812, 1, text, expect=False)


@pytest.mark.parametrize("k", [0, 1])
def test_one_pass_scan_boundaries(agh, k):
    """Occurrences at every offset around the kernel's seams: position 0 (queued by hand), the first 8 and
    last 24 bytes (k_mscan_edges), chunk / strip / supertile / range starts (position 16 of one lane is
    position 0 of the next), a partial last strip, texts below one strip, hits in every chunk."""
    rng = random.Random(k)
    # (k = 0: a 5-byte entry keeps the set on the one-pass kernel -- entries of >= 7 bytes go to the strided sweep)
    pats = [b"needlework", b"haystacks", b"abcdefgh", b"zyxwvutsrqpo"] + ([b"vwxyz"] if k == 0 else [])
    near = {0: [b"needlework", b"haystacks", b"abcdefgh"], 1: [b"needlewrk", b"haYstacks", b"abcdxefgh", b"bcdefgh"]}[k]
    for t in (b"", b"n", b"needlework", b"needlework\n", b"\nneedlework", b"xneedlework", b"abcdefgh" * 3,
              b"x" * 7 + b"abcdefgh" + b"y" * 23, b"x" * 8 + b"abcdefgh" + b"y" * 24, b"abcdefg", b"haystack"):
        _one_pass_count(agh, pats, k, t)
    for boundary in (16, 1024, 4096, 8192, 262144, 262144 + 4096):
        for tail in (0, 5, 16, 700):
            t = bytearray(b"q" * (boundary + 60 + tail))
            for i in range(41, len(t), 97):
                t[i] = 10
            for j, shift in enumerate(range(-14, 6)):
                at = boundary + shift
                w = near[j % len(near)]
                if at < 0 or at + len(w) > len(t):
                    continue
                s = bytearray(t)
                s[at:at + len(w)] = w
                for p in range(max(0, at - 2), min(len(s), at + len(w) + 2)):
                    if s[p] == 10:
                        s[p] = ord("q")
                got = _one_pass_count(agh, pats, k, bytes(s))
                assert got == 1, (boundary, tail, shift, w)
    # an occurrence in every 16-byte chunk of a range: queue A full for every supertile, queue B drained
    # in the middle of level 2
    dense = (b"abcdefgh" + b"\n" + b"r" * 7) * 40000
    assert _one_pass_count(agh, pats, k, dense) == 40000
    dense2 = (b"xabcdefgh" + b"rr\n") * 50000 + b"abcdefg"
    assert _one_pass_count(agh, pats, k, dense2) == 50000 + k      # ("abcdefg" at the end: one deletion)
    # long records: a match whose record started in another supertile / range (look-back through the text)
    longrec = b"w" * 300000 + b"abcdefgh" + b"w" * 10 + b"\n" + b"v" * 5000 + b"haystacks\n"
    assert _one_pass_count(agh, pats, k, longrec) == 2


def test_one_pass_scan_fuzz(agh):
    """Random sets, texts over small alphabets (everything is a near miss, shared prefixes, several entries
    per gram), random lengths: one-pass count == numbered count, spot-checked against the oracle."""
    rng = random.Random(2024)
    for it in range(40):
        k = rng.randint(0, 1)
        alpha = rng.choice([b"ab", b"abc", b"abcdefghijklmnopqrstuvwxyz", b"aA", b"abcd "])
        lo, hi = (8, 14) if k else (4, 15)
        nocase = rng.random() < 0.3
        palpha = alpha.replace(b" ", b"e")
        hi_len = rng.randint(lo, hi)
        # (no more patterns than a quarter of what the alphabet has at the shortest length: _rand_patterns
        # draws until it has that many DISTINCT ones)
        npat = min(rng.choice([1, 3, 30, 200]), max(1, len(set(palpha)) ** lo // 4))
        pats = _rand_patterns(rng, npat, lo, hi_len, alphabet=palpha)
        n = rng.choice([0, 1, 9, 100, 1023, 1024, 1025, 4097, 70000, 263000, 600000])
        talpha = alpha + b"\n" if rng.random() < 0.7 else alpha + b"\n\n\n\n"
        text = bytes(rng.choice(talpha) for _ in range(n))
        with agh.Query.multi(pats, nocase=nocase, k=k) as q:
            ok = q.scan_buffer(b"x" * 64, flags=agh.COUNT)[0].fused_segments == 1
        if not ok:
            continue                            # (sets of > 255 entries per gram and the like)
        got = _one_pass_count(agh, pats, k, text, nocase=nocase)
        if n <= 70000 and len(pats) <= 30:
            want = len(_approx_want(pats, k, text, nocase)) if k else O.multi_exact_count(pats, text, nocase=nocase)[0]
            assert got == want, (it, k, alpha, n)


# ---- dense sets with one error (agh_mtile.hip: BASELINE config 5 as SURVEY 8d words it) ----------
def test_record_walk_takes_dense_one_error_sets(agh):
    """1024 patterns of 4..12 bytes, k = 1: pieces of two bytes, every position a candidate, most records match.
    Count-only scans walk the records lane by lane and stop at a record's first hit; the count equals the numbered
    multi-pattern pipeline and, on a slice, the union of 1024 single-pattern oracle scans."""
    rng = random.Random(1024)
    pats = _rand_patterns(rng, 1024, 4, 12)
    base, _ = O.corpus(2048, seed=5, variants=tuple(pats[:7]), plant_period=500)       # 8 MiB
    text = base.tobytes()
    got = _one_pass_count(agh, pats, 1, text)
    nrec = text.count(b"\n")
    assert 0.5 * nrec < got <= nrec                       # (four records in five match on this alphabet)
    small = text[:200000]
    assert _one_pass_count(agh, pats, 1, small) == len(_approx_want(pats[:1024], 1, small))
    # the same set without errors, and sets the walk does not take (patterns above 14 bytes / below 4, two errors)
    with agh.Query.multi(pats, k=1) as q:
        assert q.scan_buffer(small, flags=agh.COUNT)[0].fused_segments == 1
    for ps, k in (([b"abc", b"needle"], 1), ([b"needlework", b"a" * 15], 1), ([b"needle", b"haystack"], 2)):
        with agh.Query.multi(ps, k=k) as q:
            c = q.scan_buffer(small, flags=agh.COUNT)[0]
            assert c.fused_segments == 0 and c.n_matched == q.scan_buffer(small, flags=agh.COUNT | agh.FORCE_NUMBERED)[0].n_matched


@pytest.mark.parametrize("mtile", ["2", "1", "4"])
def test_record_walk_boundaries(agh, monkeypatch, mtile):
    """Records and occurrences around the seams of the dense-set kernel (AGH_MTILE: k_mtile with 1 / 2 / 4 tiles per
    wave): a lane's 64-position word, a strip's kilobyte, a 4 KiB tile, a wave's 256 KiB range,
    the first 8 and last 24 positions (the edges kernel), records that cross several tiles with hits in each of them
    (counted once: the set of record starts), empty records, no trailing delimiter, a 2 MiB record in front of a hit
    (the give-up list), texts below 32 bytes."""
    monkeypatch.setenv("AGH_MTILE", mtile)
    pats = [b"needle", b"haystack", b"wxyz", b"abcdefghijklmn"]
    near = [b"needle", b"nedle", b"neeedle", b"haystak", b"hbystack", b"wxz", b"wxyyz", b"abcdefghijklm", b"abcdefgXijklmn"]
    for t in (b"", b"n", b"wxyz", b"wxyz\n", b"\nwxyz", b"xwxz", b"needle" * 3, b"needl", b"x" * 7 + b"needle" + b"y" * 23,
              b"x" * 8 + b"needle" + b"y" * 24, b"\n" * 100, b"wxz\n" * 10 + b"wx"):
        got = _one_pass_count(agh, pats, 1, t)
        want = _approx_want(pats, 1, t)
        assert got == len(want), t[:40]
        with agh.Query.multi(pats, k=1) as q:                   # the record LIST (numbered form: marks by record number)
            assert [(s_, e_) for s_, e_, _ in q.scan_buffer(t, cap=1000)[1]] == want, t[:40]
    for boundary in (64, 1024, 2048, 4096, 8192, 16384, 65536, 65536 + 1024, 131072, 262144):
        for tail in (0, 7, 24, 900):
            base = bytearray(b"q" * (boundary + 64 + tail))
            for i in range(53, len(base), 131):
                base[i] = 10
            for j, shift in enumerate(range(-16, 8)):
                at = boundary + shift
                w = near[j % len(near)]
                if at < 0 or at + len(w) > len(base):
                    continue
                s = bytearray(base)
                s[at:at + len(w)] = w
                for p in range(max(0, at - 2), min(len(s), at + len(w) + 2)):
                    if s[p] == 10 and not (at <= p < at + len(w)):
                        s[p] = ord("q")
                got = _one_pass_count(agh, pats, 1, bytes(s))
                assert got == len(_approx_want(pats, 1, bytes(s))) == 1, (boundary, tail, shift, w)
    # one record across five kilobytes with a hit in every one of them; its neighbours match too
    rec = (b"r" * 500 + b"needle" + b"r" * 518) * 5
    t = b"wxyz one\n" + rec + b"\n" + b"haystack two\n" + rec[:3000] + b"\n" + b"s" * 3000 + b"\nwxz"
    assert _one_pass_count(agh, pats, 1, t) == len(_approx_want(pats, 1, t)) == 5
    with agh.Query.multi(pats, k=1) as q:                       # ... and as a list: records across several tiles, by number
        res, ms = q.scan_buffer(t, cap=100)
        assert [(s_, e_) for s_, e_, _ in ms] == _approx_want(pats, 1, t) and [i_ for _, _, i_ in ms] == [0, 1, 2, 3, 5]
    # hits in every record of a long run of short records; empty records in between
    dense = (b"a needl b\n\n" + b"xx haystac yy\n") * 30000
    assert _one_pass_count(agh, pats, 1, dense) == 60000
    # a record of 2 MiB in front of a hit: its start is found after the scan (k_resolve_giveups)
    long = b"v" * (2 << 20) + b" nedle\n" + b"wxyz\n" + b"u" * 5000
    assert _one_pass_count(agh, pats, 1, long) == 2


@pytest.mark.parametrize("mtile", ["2", "1", "4"])
def test_record_walk_fuzz(agh, monkeypatch, mtile):
    """Random sets of 4..14-byte patterns over small and large alphabets (everything is a near miss, several
    entries per two-byte key, -i), random record lengths: count-only == numbered, and == the oracle's union on the
    smaller cases.  Every form of the dense-set kernel (AGH_MTILE)."""
    monkeypatch.setenv("AGH_MTILE", mtile)
    rng = random.Random(4242)
    ran = 0
    for it in range(60):
        alpha = rng.choice([b"ab", b"abc", b"abcdefghijklmnopqrstuvwxyz", b"aA", b"abcd ", b"etaoin shr"])
        nocase = rng.random() < 0.3
        palpha = alpha.replace(b" ", b"e")
        lo = rng.randint(4, 7)
        hi = rng.randint(lo, 14)
        npat = min(rng.choice([1, 3, 30, 300]), max(1, len(set(palpha)) ** lo // 4))
        pats = _rand_patterns(rng, npat, lo, hi, alphabet=palpha)
        n = rng.choice([0, 1, 9, 31, 32, 33, 100, 1023, 1024, 1025, 4095, 4096, 4097, 8200, 65535, 65536, 70000, 263000, 530000])
        talpha = alpha + b"\n" if rng.random() < 0.6 else alpha * 3 + b"\n"
        text = bytes(rng.choice(talpha) for _ in range(n))
        with agh.Query.multi(pats, nocase=nocase, k=1) as q:
            took = q.scan_buffer(b"x" * 64, flags=agh.COUNT)[0].fused_segments == 1
        if min(len(p) for p in pats) >= 8:
            continue                            # (pieces of >= 4 bytes: the filter kernels' sets, tested above)
        assert took, (it, lo, hi, npat)
        ran += 1
        got = _one_pass_count(agh, pats, 1, text, nocase=nocase)
        if n <= 70000 and len(pats) <= 30:
            want = _approx_want(pats, 1, text, nocase)
            assert got == len(want), (it, alpha, n, lo, hi)
            with agh.Query.multi(pats, nocase=nocase, k=1) as q:
                assert [(s_, e_) for s_, e_, _ in q.scan_buffer(text, cap=len(want) + 16)[1]] == want, (it, alpha, n, lo, hi)
    assert ran >= 30
