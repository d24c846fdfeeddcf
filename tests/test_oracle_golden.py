"""Pin the CPU oracle against golden vectors produced by the reference itself.

The fixtures in tests/golden/ were written by oracle/gen_golden.py, which runs the compiled
reference (oracle/_ref/agrep and oracle/_ref/ref_harness).  These tests need neither a GPU
nor /root/reference.
"""
import hashlib
import json
import os

import pytest

import _oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)["cases"]


def _delim_bytes(d):
    # agrep.c:2265-2316: "$$" / "^" mean newlines
    return d.replace("$", "\n").replace("^", "\n").encode("latin1")


def _opts_delim(opts):
    if "-d" in opts:
        return _delim_bytes(opts[opts.index("-d") + 1])
    return b"\n"


@pytest.mark.parametrize("case", _load("maskgen.json"),
                         ids=lambda c: "%s_k%d_%s" % (c["pattern"][:12], c["k"], "".join(c["opts"])))
def test_maskgen_tables_match_reference(case):
    """maskgen.c:218-266 -- every table word equals what the reference computed."""
    pat = case["pattern"].encode("latin1")
    delim = _opts_delim(case["opts"])
    M, t = O.maskgen(pat, delim, nocase="-i" in case["opts"])
    if case.get("too_long"):
        assert M == -1
        assert "pattern too long" in case["stderr"]
        return
    g = case["tables"]
    assert M == len(pat) + len(delim) + 1
    assert t.Init0 == g["Init0"]
    assert t.Init1 == g["Init1"]
    assert t.NO_ERR_MASK == g["NO_ERR_MASK"]
    assert t.endposition == g["endposition"]
    assert t.D_endpos == g["D_endpos"]
    assert t.wildmask == g["wildmask"]
    assert t.AND == g["AND"]
    assert list(t.Mask) == g["Mask"]


def test_maskgen_survey_vectors():
    """SURVEY.md 8a-a2 golden vector, independent of the JSON fixtures."""
    M, t = O.maskgen(b"approximatematch")
    assert M == 18
    assert (t.Init0, t.Init1, t.NO_ERR_MASK) == (0xfffd0000, 0xffff0001, 0xfffdffff)
    assert (t.endposition, t.D_endpos) == (1, 0x20000)
    assert t.Mask[ord("a")] == 0x8088 and t.Mask[ord("h")] == 1 and t.Mask[10] == 0x20000


def _case_text(spec):
    if spec["kind"] == "literal":
        return spec["latin1"].encode("latin1")
    text, _ = O.corpus(spec["pages"], seed=spec["seed"], variants=O.VARIANTS_C2,
                       plant_period=spec["period"])
    return text.tobytes()


def _scan_id(c):
    t = c["text"]
    tid = ("lit%d" % len(t["latin1"])) if t["kind"] == "literal" else "corp%d" % t["pages"]
    return "%s_%s_k%d_%s_%s" % (tid, c["pattern"][:8], c["k"], "".join(c["opts"]) or "cs", c["mode"])


@pytest.mark.parametrize("case", _load("scan.json"), ids=_scan_id)
def test_scan_matches_reference(case):
    """Counts and matched-record sets of the reference (both engines, both I/O modes)."""
    text = _case_text(case["text"])
    pat = case["pattern"].encode("latin1")
    k = case["k"]
    nocase = "-i" in case["opts"]
    delim = _delim_bytes(case["delim"]) if case.get("delim") else b"\n"
    if k == 0 and not nocase:
        # plain k=0 runs bm(), always case-insensitive (Q6); fixtures are lower-case text
        pass
    cap = 100000
    n_dp, r_dp = O.dp_count(pat, k, text, delim, nocase, cap)
    n_as, r_as = O.asearch(pat, k, text, delim, nocase, cap)
    n_wm, r_wm = O.wm_count(pat, k, text, delim, nocase, 64, cap)
    assert (n_as, r_as) == (n_dp, r_dp)
    assert (n_wm, r_wm) == (n_dp, r_dp)
    sgrep_path = delim == b"\n" and not nocase and k > 0 and len(pat) <= 23
    if sgrep_path:
        # the reference's own engine for this query is sgrep.c:agrep(): its literal
        # restatement reproduces the reference count INCLUDING quirk Q2, the quirk-free
        # restatement equals the Levenshtein ground truth
        n_lit, r_lit = O.sgrep_verify(pat, k, text, clean=False, cap=cap)
        n_cl, r_cl = O.sgrep_verify(pat, k, text, clean=True, cap=cap)
        assert (n_cl, r_cl) == (n_dp, r_dp)
        assert n_lit == case["count"]
        if n_lit != n_dp:
            assert n_lit < n_dp                  # Q2 only ever loses records
            r_dp = r_lit
    else:
        assert n_dp == case["count"]
    lines = b"".join(text[s:e] + b"\n" for s, e in r_dp).decode("latin1")
    if case.get("lines") is not None:
        ref_lines = case["lines"]
        if not text.endswith(b"\n") and sgrep_path and ref_lines != lines:
            # Q7: the sgrep path drops the last byte of an unterminated last record
            assert ref_lines.rstrip("\n") == lines[:-2]
        else:
            assert ref_lines == lines
    elif case.get("lines_sha256"):
        assert len(r_dp) == case["n_lines"]
        assert hashlib.sha256(lines.encode("latin1")).hexdigest() == case["lines_sha256"]


def test_quirks_are_reproduced_and_classified():
    """where the reference deviates from the Levenshtein semantic."""
    q = {c["id"]: c for c in _load("quirks.json")}
    # Q4: -c double counts a record with two far-apart occurrences (sgrep.c:1187-1193)
    c = q["Q4"]
    text = c["text"].encode("latin1")
    n_dp, _ = O.dp_count(c["pattern"].encode(), c["k"], text)
    assert n_dp == 1 and c["asearch_count"] == 1 and c["sgrep_count"] == 2
    assert c["sgrep_lines"].count("\n") == 1
    # Q2: literal restatement of sgrep.c:1196-1201 loses the record after a matched one
    c = q["Q2"]
    text = c["text"].encode("latin1")
    pat = c["pattern"].encode()
    n_dp, _ = O.dp_count(pat, c["k"], text)
    assert n_dp == 2 and c["asearch_count"] == 2 and c["sgrep_count"] == 1
    assert O.sgrep_verify(pat, c["k"], text, clean=False)[0] == 1
    assert O.sgrep_verify(pat, c["k"], text, clean=True)[0] == 2
    # Q6: bm() folds case without -i (sgrep.c:226-236)
    c = q["Q6"]
    assert c["sgrep_count"] == 2
    assert O.dp_count(b"hello", 0, c["text"].encode("latin1"))[0] == 1
    assert O.dp_count(b"hello", 0, c["text"].encode("latin1"), nocase=True)[0] == 2


@pytest.mark.parametrize("case", _load("costs.json"),
                         ids=lambda c: "%s_k%d_I%dS%dD%d" % (c["text"]["kind"][:4], c["k"], *c["costs"]))
def test_weighted_costs_match_reference(case):
    """asearch1.c (-I# -S# -D#): counts and printed records of the reference CLI."""
    text = _case_text(case["text"])
    pat = case["pattern"].encode("latin1")
    n, recs = O.asearch_costs(pat, case["k"], tuple(case["costs"]), text, cap=100000)
    assert n == case["count"]
    lines = b"".join(text[s:e] + b"\n" for s, e in recs).decode("latin1")
    assert len(recs) == case["n_lines"]
    assert hashlib.sha256(lines.encode("latin1")).hexdigest() == case["lines_sha256"]
    if tuple(case["costs"]) == (1, 1, 1):
        assert (n, recs) == O.asearch(pat, case["k"], text, cap=100000)


@pytest.mark.parametrize("case", _load("exact_segments.json"), ids=lambda c: c["pattern"])
def test_exact_segments_match_reference(case):
    """<...> segments: the oracle automaton driven by the reference's own tables reproduces the
    reference count (NO_ERR_MASK forbids error transitions into those positions)."""
    text = _case_text(case["text"])
    m = len(case["pattern"]) - 2                      # '<' and '>' are not positions
    t = O.tables_from_golden(case["tables"], m + 2)
    assert O.asearch_tables(t, case["k"], text)[0] == case["count"]


def _lang_M(t):
    return t["D_endpos"].bit_length()           # D_endpos = 1 << (M - 1) for a one-byte delimiter


def _lang_text(case):
    return _case_text(case["text"]) + case["extra_latin1"].encode("latin1")


@pytest.mark.parametrize("case", _load("pattern_language.json"),
                         ids=lambda c: "%s_k%d_%s" % (c["pattern"], c["k"], "".join(c["opts"]) or "plain"))
def test_pattern_language_tables_match_reference(case):
    """Character classes, -w and -x live entirely in maskgen's tables (preproce.c:148-175,
    maskgen.c:86-170): the oracle automaton driven by those tables gives the reference count."""
    t = O.tables_from_golden(case["tables"], _lang_M(case["tables"]))
    assert O.asearch_tables(t, case["k"], _lang_text(case))[0] == case["count"]


def lang_delims_text(case):
    """Text of a pattern_language_delims.json case (oracle/gen_golden.py:gen_pattern_language_delims)."""
    spec = case["text"]
    text, _ = O.corpus(spec["pages"], seed=spec["seed"], variants=O.VARIANTS_C2, plant_period=spec["period"])
    delim = case["delim_latin1"].encode("latin1")
    tb = (text.tobytes() + spec["words_latin1"].encode("latin1")).replace(b"\n", delim)
    if delim == b"xy":
        tb = tb.replace(b"xy", b"xY", 7).replace(b"xy", b"XY", 5)
    return tb, delim


@pytest.mark.parametrize("case", _load("pattern_language_delims.json"),
                         ids=lambda c: "%s_k%d_%s" % (c["pattern"], c["k"], "".join(c["opts"]).replace("\r\n", "CRLF")))
def test_pattern_language_with_multi_byte_delimiters_matches_reference(case):
    """'#', ';' and ',' under -d delimiters of several bytes (also letters under -i): the oracle automaton on the
    reference's tables -- delimiter positions in front, D_Mask over all of them (asearch.c:54-57) -- gives
    the reference's count, including where the delimiter's own bytes take part in an occurrence (Q11)."""
    text, delim = lang_delims_text(case)
    tb = case["tables"]
    t = O.tables_from_golden(tb, _lang_M(tb) + len(delim) - 1, dlen=len(delim))
    assert O.asearch_tables(t, case["k"], text, delim=delim)[0] == case["count"]
