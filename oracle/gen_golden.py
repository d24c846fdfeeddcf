#!/usr/bin/env python3
"""Generate tests/golden/*.json by RUNNING THE REFERENCE ITSELF (oracle/_ref).

TEST INFRASTRUCTURE.  Needs `make -C oracle ref` (i.e. /root/reference present); the
fixtures it writes are committed so that the GPU box / CI never needs the reference.

  maskgen.json : the reference's own query tables (Mask[256], Init[0], Init1, NO_ERR_MASK,
                 endposition, D_endpos) for a list of (pattern, options) -- maskgen.c:218-266
  scan.json    : for (text, pattern, k, options): the reference's -c count and, for the
                 newline delimiter, the matched records it prints; both engines
                 (sgrep.c:agrep() for plain k>0 queries, asearch.c for -i queries) and both
                 I/O modes (file mode via the CLI, memory mode via memagrep()).
  quirks.json  : small inputs on which the reference deviates from the Levenshtein
                 semantic, with what it prints.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
AGREP = os.path.join(REF, "agrep")
HARNESS = os.path.join(REF, "ref_harness")
OUT = os.path.join(ROOT, "tests", "golden")


def run(cmd, stdin=None, timeout=None):
    try:
        p = subprocess.run(cmd, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout,
                           stdin=subprocess.DEVNULL if stdin is None else None)
    except subprocess.TimeoutExpired:
        return -9, b"", b"(the reference did not return)"
    return p.returncode, p.stdout, p.stderr


def gen_maskgen():
    cases = []
    specs = [
        # -n keeps a plain literal off the SGREP path (checksg.c:132) so maskgen runs
        ("abcab", 1, ["-n"]), ("abcab", 1, ["-i"]),
        ("approximatematch", 2, ["-n"]), ("approximatematch", 2, ["-i"]),
        ("MiXed", 1, ["-n"]),
        ("ApproXimateMatch", 2, ["-i"]),
        ("homogenos", 2, ["-i"]), ("a", 0, ["-i"]), ("ab", 1, ["-i"]),
        ("abcdefghijklmnopqrstuvwxyzabc", 3, ["-i"]),          # m = 29, the maximum
        ("matching", 1, ["-i", "-d", "From "]),
        ("matching", 2, ["-i", "-d", "$$"]),
        ("hello world", 4, ["-i"]),
        ("zzzyzzzy", 7, ["-i"]), ("abcdefghijkl", 8, ["-i"]),
    ]
    for pat, k, opts in specs:
        if "-n" not in opts:
            opts = opts + ["-n"]
        args = [HARNESS, "tables"] + (["-%d" % k] if k else []) + opts + [pat]
        rc, out, err = run(args)
        assert rc == 0, (args, err)
        t = json.loads(out)
        # the reference runs maskgen only on the non-SGREP path (agrep.c:3181-3193)
        assert t["SGREP"] == 0, (pat, opts)
        cases.append({"pattern": pat, "k": k, "opts": opts, "tables": t})
    # too long: m = 30 with newline delimiter is rejected (maskgen.c:201-208)
    rc, out, err = run([HARNESS, "tables", "-2", "-i", "-n", "abcdefghijklmnopqrstuvwxyzabcd"])
    cases.append({"pattern": "abcdefghijklmnopqrstuvwxyzabcd", "k": 2, "opts": ["-i"],
                  "too_long": True, "stderr": err.decode("latin1")})
    with open(os.path.join(OUT, "maskgen.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "cases": cases}, f, indent=0)
    print("maskgen.json:", len(cases), "cases")


def ref_scan(text, pat, k, opts, mode):
    """-> (count, lines or None)"""
    with tempfile.NamedTemporaryFile(suffix=".txt", delete=False) as tf:
        tf.write(text)
        path = tf.name
    try:
        kopt = ["-%d" % k] if k else []
        if mode == "file":
            rc, out, err = run([AGREP, "-V0"] + kopt + opts + ["-c", pat, path])
            cnt = int(out.split()[0]) if out.strip() else 0
            rc, out, err = run([AGREP, "-V0"] + kopt + opts + [pat, path])
            lines = out.decode("latin1")
        else:
            rc, out, err = run([HARNESS, "count", path] + kopt + opts + [pat])
            cnt = int(out.split()[0])
            rc, out, err = run([HARNESS, "lines", path] + kopt + opts + [pat])
            lines = out.decode("latin1")
        return cnt, lines
    finally:
        os.unlink(path)


def gen_scan():
    cases = []
    pat = O.PATTERN_C2
    # (a) generator corpora, dense planting so every variant shows up
    for pages, period, seed in [(16, 20, 12345), (64, 50, 7), (24, 3, 99)]:
        text, planted = O.corpus(pages, seed=seed, variants=O.VARIANTS_C2, plant_period=period)
        tb = text.tobytes()
        for k in (0, 1, 2, 3):
            for opts in ([], ["-i"]):
                for mode in ("file", "mem"):
                    if k == 0 and mode == "mem":
                        continue
                    cnt, lines = ref_scan(tb, pat.decode(), k, opts, mode)
                    cases.append({"text": {"kind": "corpus", "pages": pages, "seed": seed,
                                           "period": period},
                                  "pattern": pat.decode(), "k": k, "opts": opts, "mode": mode,
                                  "count": cnt, "lines": None, "n_lines": lines.count("\n"),
                                  "lines_sha256": hashlib.sha256(lines.encode("latin1")).hexdigest()})
    # (b) hand-written edge cases (newline delimiter)
    lit = [
        (b"", "abc", 1), (b"\n", "abc", 1), (b"\n\n\n", "abc", 1),
        (b"abc", "abc", 0), (b"abc\n", "abc", 0), (b"xabcx", "abd", 1),
        (b"hello world\nhomogeneous\nfoo\nhomogenos", "homogenos", 2),
        (b"Massachusetts\nmassive\nMassechusets\n", "Massechusets", 2),
        (b"aaaaaaaaaaaaaaaaaaaaaaaaaaaaaa\nbbbb\n", "aaaaaaaa", 3),
        (b"ab\nabcdefgh\nabcdefg\nbcdefgh\nabcXefgh\nabcdeXXfgh\n", "abcdefgh", 1),
        (b"x" * 5000 + b"needle" + b"y" * 5000 + b"\nshort\n", "needle", 1),
        (b"first line has no newline at the end but matcZ", "match", 1),
    ]
    for text, p, k in lit:
        for opts in ([], ["-i"]):
            if k == 0 and not opts:
                # plain k=0 goes to bm() which is always case-insensitive (Q6); the inputs
                # here are lower-case only so it is still comparable
                pass
            for mode in ("file", "mem"):
                if mode == "mem" and (not text.endswith(b"\n")):
                    continue  # memory mode never closes an unterminated record (asearch.c:326-345)
                cnt, lines = ref_scan(text, p, k, opts, mode)
                cases.append({"text": {"kind": "literal", "latin1": text.decode("latin1")},
                              "pattern": p, "k": k, "opts": opts, "mode": mode,
                              "count": cnt, "lines": lines})
    # (c) custom delimiters: counts only (the printed form depends on -t / OUTTAIL)
    mbox = (b"From alice\nsubject: approximate matching\nbody\n"
            b"From bob\nnothing here\n\nFrom carol\napproximatematch is here\n"
            b"From dave\naproximatemach twice removed\n")
    para = b"para one\nline\n\npara two approximatematch\n\n\npara three aproximatematc\nx\n\n"
    for text, dl in ((mbox, "From "), (para, "$$")):
        for k in (0, 1, 2):
            for opts in ([], ["-i"]):
                rc, out, err = run([AGREP, "-V0"] + (["-%d" % k] if k else []) + opts +
                                   ["-d", dl, "-c", pat.decode(), "/dev/stdin"], stdin=text)
                cnt = int(out.split()[0]) if out.strip() else 0
                cases.append({"text": {"kind": "literal", "latin1": text.decode("latin1")},
                              "pattern": pat.decode(), "k": k, "opts": opts, "mode": "file",
                              "delim": dl, "count": cnt, "lines": None})
    with open(os.path.join(OUT, "scan.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "cases": cases}, f, indent=0)
    print("scan.json:", len(cases), "cases")


def gen_costs():
    """-I# -S# -D# (asearch1.c): counts and lines from the reference CLI."""
    cases = []
    pat = O.PATTERN_C2.decode()
    text, _ = O.corpus(24, seed=99, variants=O.VARIANTS_C2, plant_period=3)
    tb = text.tobytes()
    extra = (b"approximatematch\naproximatematch\napproxximatematch\napproxXmatematch\n"
             b"apprximatemtch\napproximatematchh\nappproximatematch\nfoo\n")
    for body, kind in ((tb, {"kind": "corpus", "pages": 24, "seed": 99, "period": 3}),
                       (extra, {"kind": "literal", "latin1": extra.decode("latin1")})):
        for k, costs in ((2, (2, 1, 1)), (2, (1, 2, 1)), (2, (1, 1, 2)), (2, (3, 1, 3)), (1, (1, 2, 1)),
                         (2, (1, 2, 1)), (3, (2, 2, 1)), (3, (1, 3, 2)), (4, (2, 3, 2)), (2, (1, 1, 1)),
                         (2, (9, 9, 9))):
            opts = ["-I%d" % costs[0], "-S%d" % costs[1], "-D%d" % costs[2]]
            cnt, lines = ref_scan(body, pat, k, opts, "file")
            cases.append({"text": kind, "pattern": pat, "k": k, "costs": list(costs), "count": cnt,
                          "n_lines": lines.count("\n"),
                          "lines_sha256": hashlib.sha256(lines.encode("latin1")).hexdigest()})
    with open(os.path.join(OUT, "costs.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "cases": cases}, f, indent=0)
    print("costs.json:", len(cases), "cases", [c["count"] for c in cases])


def gen_exact_segments():
    """<...> exact segments (NO_ERR_MASK, maskgen.c:80-95): tables + counts from the reference."""
    cases = []
    text, _ = O.corpus(24, seed=99, variants=O.VARIANTS_C2, plant_period=3)
    tb = text.tobytes()
    for pat, k in (("<appro>ximatematch", 2), ("approxi<matematch>", 2), ("ap<proxim>atematch", 1),
                   ("<approximatematch>", 2)):
        rc, out, err = run([HARNESS, "tables", "-%d" % k, "-n", pat])
        t = json.loads(out)
        cnt, lines = ref_scan(tb, pat, k, ["-n"], "file")
        cases.append({"pattern": pat, "k": k, "tables": t, "count": cnt,
                      "text": {"kind": "corpus", "pages": 24, "seed": 99, "period": 3}})
    with open(os.path.join(OUT, "exact_segments.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "cases": cases}, f, indent=0)
    print("exact_segments.json:", [(c["pattern"], c["count"], hex(c["tables"]["NO_ERR_MASK"])) for c in cases])


def gen_pattern_language():
    """Non-literal patterns whose whole meaning is in maskgen's tables: character classes,
    -w word guards, -x whole-line guards.  Tables + counts from the reference."""
    cases = []
    text, _ = O.corpus(24, seed=99, variants=O.VARIANTS_C2, plant_period=3)
    words = (b"the car is red\ncars are fast\na scar\ncar\ncharacter\nmy car.\ncat\n"
             b"approximatematch\nxapproximatematch\napproximatematch x\n")
    tb = text.tobytes() + words
    for pat, k, opts in (("appr[ox]ximatematch", 2, []), ("approx[a-m]matematch", 1, []),
                         ("[^b-z]pproximatematch", 2, []), ("car", 1, ["-w"]), ("car", 0, ["-w"]),
                         ("approximatematch", 2, ["-w"]), ("approximatematch", 1, ["-x"]),
                         ("car", 0, ["-x"]),
                         ("approx#match", 0, []), ("approx#match", 1, []), ("appr#mate#ch", 2, []),
                         ("approxi;matematch", 0, []), ("approxi;matematch", 1, []),
                         ("matematch;approx", 2, []), ("cars;fast", 0, []),
                         ("approxi,xyzzyq", 0, []), ("aproxi,matemmat", 1, []), ("scar,cat", 0, []),
                         ("a#h", 0, ["-w"]), ("car;red", 1, []),
                         ("^appro", 0, []), ("^aproxi", 1, []), ("tematch$", 1, []), ("^car$", 0, []),
                         ("^cars", 1, []), ("fast$", 0, [])):
        kopt = ["-%d" % k] if k else []
        rc, out, err = run([HARNESS, "tables"] + kopt + ["-n"] + opts + [pat])
        t = json.loads(out)
        cnt, lines = ref_scan(tb, pat, k, ["-n"] + opts, "file")
        cases.append({"pattern": pat, "k": k, "opts": opts, "tables": t, "count": cnt,
                      "extra_latin1": words.decode("latin1"),
                      "text": {"kind": "corpus", "pages": 24, "seed": 99, "period": 3}})
    with open(os.path.join(OUT, "pattern_language.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "cases": cases}, f, indent=0)
    print("pattern_language.json:", [(c["pattern"], c["opts"], c["count"], c["tables"]["ret"]) for c in cases])


def gen_pattern_language_delims():
    """The table-engine patterns ('#', ';', ',') under delimiters of several bytes (-d): tables and
    counts from the reference.  `delim` is what separates the records in the text ('$' and '^' in the
    option mean newline, bitap.c:92-94)."""
    cases = []
    text, _ = O.corpus(8, seed=77, variants=O.VARIANTS_C2, plant_period=3)
    words = (b"the car is red\ncars are fast\na scar\ncar\ncharacter\nmy car.\ncat\n"
             b"approximatematch\nxapproxQmatch\napprox match x\nred car\nfast cars\n")
    base = text.tobytes() + words
    for dopt, delim, more in ((";;", b";;", []), ("\r\n", b"\r\n", []), ("$$", b"\n\n", []), ("; ", b"; ", []),
                              ("XY", b"xy", ["-i"]), (":::", b":::", []), ("@@@@", b"@@@@", [])):
        tb = base.replace(b"\n", delim)
        if delim == b"xy":                      # -i: the delimiter's letters in both cases
            tb = tb.replace(b"xy", b"xY", 7).replace(b"xy", b"XY", 5)
        for pat, k in (("approx#match", 0), ("approx#match", 1), ("appr#mate#ch", 2), ("approxi;matematch", 1),
                       ("cars;fast", 0), ("scar,cat", 0), ("aproxi,matemmat", 1), ("car;red", 1)):
            opts = more + ["-d", dopt]
            kopt = ["-%d" % k] if k else []
            rc, out, err = run([HARNESS, "tables"] + kopt + ["-n"] + opts + [pat])
            t = json.loads(out)
            cnt, lines = ref_scan(tb, pat, k, ["-n"] + opts, "file")
            cases.append({"pattern": pat, "k": k, "opts": opts, "delim_latin1": delim.decode("latin1"),
                          "nocase": "-i" in more, "tables": t, "count": cnt,
                          "text": {"kind": "corpus+words, newlines replaced by the delimiter", "pages": 8, "seed": 77,
                                   "period": 3, "words_latin1": words.decode("latin1")}})
    with open(os.path.join(OUT, "pattern_language_delims.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "cases": cases}, f, indent=0)
    print("pattern_language_delims.json:", [(c["pattern"], c["opts"], c["count"]) for c in cases])


def gen_quirks():
    q = []
    pat = "approximatematch"
    # Q4: two occurrences far apart in one record: -c counts 2, one line printed (sgrep path)
    rec = b"approxXmatematch" + b" filler" * 20 + b"appQoximRtematch\n"
    cnt, lines = ref_scan(b"zzz\n" + rec + b"yyy\n", pat, 2, [], "file")
    cnt_i, lines_i = ref_scan(b"zzz\n" + rec + b"yyy\n", pat, 2, ["-i"], "file")
    q.append({"id": "Q4", "text": (b"zzz\n" + rec + b"yyy\n").decode("latin1"), "pattern": pat,
              "k": 2, "sgrep_count": cnt, "sgrep_lines": lines, "asearch_count": cnt_i})
    # Q2: record right after a matched record needs a leading deletion
    text = b"xx approximatematch yy\npproximatematch trailing words here\nzz\n"
    cnt, lines = ref_scan(text, pat, 1, [], "file")
    cnt_i, lines_i = ref_scan(text, pat, 1, ["-i"], "file")
    q.append({"id": "Q2", "text": text.decode("latin1"), "pattern": pat, "k": 1,
              "sgrep_count": cnt, "sgrep_lines": lines, "asearch_count": cnt_i,
              "asearch_lines": lines_i})
    # Q6: plain k=0 (bm) is always case-insensitive
    text = b"Hello World\nhello world\n"
    cnt, lines = ref_scan(text, "hello", 0, [], "file")
    q.append({"id": "Q6", "text": text.decode("latin1"), "pattern": "hello", "k": 0,
              "sgrep_count": cnt, "sgrep_lines": lines})
    with open(os.path.join(OUT, "quirks.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "cases": q}, f, indent=0)
    print("quirks.json:", [(c["id"], c["sgrep_count"], c.get("asearch_count")) for c in q])
def gen_pattern_compiler():
    """Tables only, for the library's own pattern compiler (agrep_amd/csrc/agh_pattern.cpp): the non-regex pattern
    language as preprocess() + maskgen() see it -- '.', '#', escapes, <exact> segments, classes with ranges,
    negation and escaped members, ';' / ',' lists, ^ / $ anchors -- alone and under -i / -w / -x / -d, from a fixed
    list and from a seeded random walk over the same tokens.  A case the reference turns down carries "ret" != 0
    or "failed": the compiler has to refuse it too."""
    import random
    fixed = [("a.c", []), ("a\\.c", []), ("a\\#c", []), ("ab#cd", []), ("<abc>de", []), ("ab<cd>ef", []), ("<ab>c<de>", []),
             ("[a-c]xyz", []), ("x[^a-c]yz", []), ("xy[a\\]b]z", []), ("xy[a\\-c]z", []), ("[0-9][0-9]:[0-9][0-9]", []),
             ("[A-Z]bc", ["-i"]), ("a[b-d]E", ["-i"]), ("x[^A-C]y", ["-i"]), ("Ab.Cd", ["-i"]), ("<Ab>cd", ["-i"]),
             ("c[ao]r", ["-w"]), ("c.r", ["-w"]), ("c[ao]r", ["-x"]), ("ca#r", ["-x"]), ("<ca>r", ["-w"]),
             ("a.c", ["-d", "From "]), ("[a-c]#z", ["-d", "$$"]), ("<ab>c.", ["-i", "-d", "XY"]),
             ("ab;c.d", []), ("a[bc];d#e", []), ("ab,c[de]", []), ("<ab>,cd", []), ("^a.c", []), ("a[bc]$", []), ("^<ab>c$", []),
             ("a\\<b", []), ("a\\[b", []), ("a\\^b\\$", []), ("a\\;b\\,c", []), ("ab\\*c", []), ("a\\|b", []), ("\\(ab\\)", []),
             ("a<bc", []), ("ab>c", []), ("ab[cd", []), ("a;b,c", []), ("a" * 29, []), ("a" * 30, []), ("a" * 25, ["-d", "From "]),
             ("a.b#c[d-f]<gh>ijklmnopqrstuvwxy", []), ("a" * 28, ["-w"]), ("a" * 27 + ".", ["-x"])]
    rng = random.Random(2026)
    toks = ["a", "b", "c", "Q", "z", "0", "7", " ", ".", "#", "\\.", "\\#", "\\\\", "[a-c]", "[^x-z]", "[abQ]", "[0-9a-f]", "<ab>", "<Qz0>",
            "[a\\]]", "[A-C]", "-", "_", "\\-"]
    walk = []
    for i in range(110):
        n = rng.randint(1, 9)
        body = "".join(rng.choice(toks) for _ in range(n))
        if body[0] == "-":                       # (the harness would read it as an option)
            body = "a" + body
        r = rng.random()
        if r < 0.12:
            body += ";" + "".join(rng.choice(toks[:8]) for _ in range(rng.randint(1, 4)))
        elif r < 0.2:
            body += "," + "".join(rng.choice(toks[:8]) for _ in range(rng.randint(1, 4)))
        elif r < 0.26:
            body = "^" + body
        elif r < 0.32:
            body += "$"
        opts = []
        if rng.random() < 0.35:
            opts.append("-i")
        r = rng.random()
        if r < 0.15:
            opts.append("-w")
        elif r < 0.3:
            opts.append("-x")
        if rng.random() < 0.2:
            opts += ["-d", rng.choice(["$$", ";;", "From ", "@@@"])]
        walk.append((body, opts))
    cases = []
    for pat, opts in fixed + walk:
        rc, out, err = run([HARNESS, "tables", "-n"] + opts + [pat], timeout=10)
        if rc == -9:
            print("  (the reference hangs on", repr(pat), opts, "-- left out)")
            continue
        case = {"pattern": pat, "opts": opts}
        if "-d" in opts:
            d = opts[opts.index("-d") + 1]
            d = d.replace("$", "\n").replace("^", "\n")
            if "-i" in opts:
                d = d.lower()
            case["delim_latin1"] = d
        try:
            t = json.loads(out)
        except ValueError:
            t = None
        if rc != 0 or t is None or t["ret"] < 0:          # ("ret" >= 0: records the harness's own text matched)
            case["failed"] = True
            case["stderr"] = err.decode("latin1")[:200]
        elif t["SGREP"]:
            case["sgrep"] = True          # the reference never ran maskgen on it (agrep.c:3181-3193)
        else:
            case["tables"] = t
        cases.append(case)
    with open(os.path.join(OUT, "pattern_compiler.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "cases": cases}, f, indent=0)
    print("pattern_compiler.json:", len(cases), "cases,", sum("tables" in c for c in cases), "with tables,",
          sum("failed" in c for c in cases), "refused by the reference")


if __name__ == "__main__":
    assert O.have_ref(), "run `make -C oracle ref` first (needs /root/reference)"
    os.makedirs(OUT, exist_ok=True)
    gen_maskgen()
    gen_scan()
    gen_quirks()
    gen_costs()
    gen_exact_segments()
    gen_pattern_language()
    gen_pattern_language_delims()
    gen_pattern_compiler()
