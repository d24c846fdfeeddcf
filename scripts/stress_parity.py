"""Randomised parity stress: random patterns / k / -i / delimiters / texts, every device engine
against the oracle.  usage: stress_parity.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
os.environ["AGH_FUSED_MIN_MB"] = "0"          # count-only scans: the fused kernel at every size ...
os.environ["AGH_ENV_LIVE"] = "1"              # (switches are flipped between scans of one query)
import numpy as np
import agrep_amd as A
import _oracle as O

import json
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "pattern_language.json")))["cases"]
GOLD_D = json.load(open(os.path.join(ROOT, "tests", "golden", "pattern_language_delims.json")))["cases"]


def table_case():
    """'#' / ';' / ',' / -p patterns on the reference's own tables (table engine, fast form and
    k_tablescan, with and without edit costs) against the oracle's asearch on the same tables."""
    global fails
    multi_d = rng.random() < 0.4                # ... under a delimiter of several bytes (round 3)
    case = rng.choice(GOLD_D if multi_d else GOLD)
    tb = case["tables"]
    delim = case["delim_latin1"].encode("latin1") if multi_d else b"\n"
    dopt = case["opts"][case["opts"].index("-d") + 1].encode("latin1") if multi_d else b"\n"
    M = tb["D_endpos"].bit_length() + len(delim) - 1
    letters = bytes(sorted(set(c for c in case["pattern"].encode() if chr(c).isalnum()))) or b"a"
    if multi_d:
        dset = delim + (delim.upper() if case.get("nocase") else b"")
        alpha = letters + rng.choice([dset, dset + b" ", dset + dset + b"xyz "])
    else:
        alpha = letters + rng.choice([b"\n", b" \n", b"xyz \n", b"\n\n"])
    n = rng.choice([0, 1, 17, 300, 4095, 4096, 4097, 70000, 262144, 300000, 1 << 20])
    text = bytearray(rng.choice(alpha) for _ in range(min(n, 8192)))
    if n > 8192:
        arr = np.tile(np.frombuffer(bytes(text), dtype=np.uint8), n // 8192 + 1)[:n].copy()
        g = np.random.default_rng(rng.randrange(1 << 30))
        idx = g.integers(0, n, size=n // 20)
        arr[idx] = np.frombuffer(alpha, dtype=np.uint8)[g.integers(0, len(alpha), size=len(idx))]
        if rng.random() < 0.3:                  # a few very long records
            lo = rng.randrange(0, n // 2)
            seg = arr[lo:lo + rng.choice([5000, 70000, 300000])]
            seg[seg == delim[-1]] = letters[0]
        text = bytearray(arr.tobytes())
    text = bytes(text)
    costs = None
    if case["k"] > 0 and rng.random() < 0.4:
        costs = (rng.randint(1, 2), rng.randint(1, 2), rng.randint(1, 2))
    if not text and multi_d:
        return                                  # (empty text: only the appended delimiter could match, Q11)
    ot = O.tables_from_golden(tb, M, dlen=len(delim))
    want = (O.asearch_tables_costs(ot, case["k"], costs, text, delim=delim, cap=400000) if costs
            else O.asearch_tables(ot, case["k"], text, delim=delim, cap=400000))
    q = A.Query.from_maskgen(tb["Mask"], tb["Init0"], tb["Init1"], tb["NO_ERR_MASK"], tb["endposition"],
                             tb["D_endpos"], M, dopt, case["k"], tb["AND"])
    try:
        if costs:
            q.set_costs(*costs)
        for fast in ("1", "0"):
            os.environ["AGH_FS_FAST"] = fast
            r1, ms = q.scan_buffer(text, cap=400000)
            r2, _ = q.scan_buffer(text, flags=A.COUNT)
            r3, _ = q.scan_buffer(text, flags=A.COUNT | A.FORCE_NUMBERED)
            got = (r1.n_matched, [(s_, e_) for s_, e_, _ in ms])
            if got != want or not (r2.n_matched == r3.n_matched == want[0]):
                fails += 1
                print("MISMATCH table", case["pattern"], case["opts"], "delim", delim, "k", case["k"], "costs", costs, "fast", fast, "n", len(text),
                      "want", want[0], "got", r1.n_matched, r2.n_matched, r3.n_matched, "seed", seed, "case", n_cases, flush=True)
    finally:
        os.environ.pop("AGH_FS_FAST", None)
        q.close()


t_end = time.time() + budget
n_cases = 0
fails = 0
ALPH = [b"ab", b"abc", b"acgt", b"abcdefghij", b"abcdefghijklmnopqrstuvwxyz", b"aAbB \n", b"ab\n"]
while time.time() < t_end:
    alpha = rng.choice(ALPH)
    m = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24, 29])
    k = rng.choice([0, 0, 1, 1, 2, 2, 3, 4, 5])
    if k >= m:
        k = m - 1
    letters = bytes(c for c in alpha if c != 10) or b"a"
    pat = bytes(rng.choice(letters) for _ in range(m))
    nocase = rng.random() < 0.3
    kind = rng.choice(["single"] * 5 + ["costs", "costs", "mbdelim", "wide", "multi", "multi", "multi", "table", "table"])
    if kind == "table":
        table_case()
        n_cases += 1
        continue
    delim = b"\n" if rng.random() < 0.8 else bytes([rng.choice(b";|\t")])
    if kind == "mbdelim" or (kind == "multi" and rng.random() < 0.25):
        delim = rng.choice([b"\n\n", b";;", b"\r\n", b"ab\n", b"$$"]) if kind == "mbdelim" else rng.choice([b"\n\n", b";;", b"\r\n", b"$$"])
        alpha = alpha + delim
    if kind == "wide":
        m = rng.choice([30, 33, 40, 48, 64])
        k = min(k, 4)
        pat = bytes(rng.choice(letters) for _ in range(m))
        alpha = b"abcdefgh\n" if len(letters) < 4 else alpha
        letters = bytes(c for c in alpha if c != 10)
        pat = bytes(rng.choice(letters) for _ in range(m))
    if nocase and len(delim) == 1 and chr(delim[0]).isalpha():
        nocase = False                  # -i with a letter as single-byte delimiter: rejected by design
    if kind == "mbdelim" and len(pat) + len(delim) > 29:
        pat = pat[:20]
        k = min(k, len(pat) - 1)
    if (kind != "mbdelim" and any(c in pat for c in delim)) or 10 in pat:
        continue
    n = rng.choice([0, 1, 5, 100, 1023, 1024, 1025, 4096, 70000, 262144, 262145, 600000])
    if rng.random() < 0.12 and len(delim) == 1:         # several 1 MiB device segments
        n = rng.choice([2500000, 3 << 20])
        os.environ["AGH_SEG_MAX_MB"] = "1"
    else:
        os.environ.pop("AGH_SEG_MAX_MB", None)
    arr = np.frombuffer(bytes(rng.choice(alpha) for _ in range(min(n, 4096))), dtype=np.uint8)
    if n > 4096:
        arr = np.tile(arr, n // 4096 + 1)[:n].copy()
        # break the periodicity: random edits and delimiters
        idx = np.random.default_rng(rng.randrange(1 << 30)).integers(0, n, size=n // 50)
        arr[idx] = np.frombuffer(bytes(rng.choice(alpha) for _ in range(len(idx))), dtype=np.uint8) if len(idx) < 20000 else arr[idx[::-1]]
        arr[np.random.default_rng(rng.randrange(1 << 30)).integers(0, n, size=n // rng.choice([30, 80, 300]))] = delim[0]
    text = bytearray(arr.tobytes())
    # plant near-occurrences
    for _ in range(rng.randint(0, 20)):
        if len(text) <= m + 2:
            break
        v = bytearray(pat)
        for _e in range(rng.randint(0, k + 1)):
            op = rng.randint(0, 2)
            at = rng.randrange(len(v)) if v else 0
            if op == 0 and len(v) > 1:
                del v[at]
            elif op == 1:
                v.insert(at, rng.choice(letters))
            elif v:
                v[at] = rng.choice(letters)
        at = rng.randrange(0, len(text) - len(v))
        text[at:at + len(v)] = v
    if rng.random() < 0.5 and text and text[-1] != delim[0]:
        text += delim
    text = bytes(text)
    if not text and len(delim) > 1:
        continue                        # empty text: only the appended delimiter could match (Q11)
    if kind == "multi":
        npat = rng.choice([1, 2, 5, 20])
        pats = sorted({bytes(rng.choice(letters) for _ in range(rng.randint(max(2, k + 1), 10))) for _ in range(npat)})
        if any(delim[-1] in x for x in pats) or len(text) > 300000:
            continue
        recs = set()
        for x in pats:
            recs.update(O.asearch(x, k, text, delim=delim, nocase=nocase, cap=300000)[1])
        want_m = sorted(recs)
        try:
            with A.Query.multi(pats, nocase=nocase, delim=delim, k=k) as q:
                r1, ms = q.scan_buffer(text, cap=300000)
                r2, _ = q.scan_buffer(text, flags=A.COUNT)
                r3, _ = q.scan_buffer(text, flags=A.COUNT | A.FORCE_NUMBERED)
            got_m = [(s_, e_) for s_, e_, _ in ms]
            if got_m != want_m or not (r1.n_matched == r2.n_matched == r3.n_matched == len(want_m)):
                fails += 1
                print("MISMATCH multi", pats, "k", k, nocase, delim, len(text), len(want_m), r1.n_matched, r2.n_matched, r3.n_matched, flush=True)
        except A.AghError as e:
            print("ERROR multi", e, pats, k, flush=True)
            fails += 1
        n_cases += 1
        continue
    costs = None
    if kind == "costs" and len(pat) + len(delim) <= 30:
        costs = (rng.randint(1, 3), rng.randint(1, 3), rng.randint(1, 3))
        want = O.asearch_costs(pat, k, costs, text, delim=delim, nocase=nocase, cap=300000)
    elif len(pat) + len(delim) <= 30:
        want = O.asearch(pat, k, text, delim=delim, nocase=nocase, cap=300000)
    else:
        want = O.wm_count(pat, k, text, delim=delim, nocase=nocase, word_bits=64, cap=300000)
    try:
        with A.Query(pat, k, nocase=nocase, delim=delim) as q:
            if costs:
                q.set_costs(*costs)
            got = {}
            for lab, fl, cap in (("default", 0, 300000), ("fullscan", A.FORCE_FULLSCAN, 300000),
                                 ("lean", A.COUNT, 0), ("lean, two kernels", A.COUNT, 0),
                                 ("numbered", A.COUNT | A.FORCE_NUMBERED, 0)):
                os.environ["AGH_FUSED"] = "0" if lab == "lean, two kernels" else "1"   # ... and its two-kernel form
                res, ms = q.scan_buffer(text, flags=fl, cap=cap)
                full_list = cap and want[0] <= cap          # else the device list is a truncated subset
                got[lab] = (res.n_matched, [(s, e) for s, e, _ in ms]) if full_list else (res.n_matched, want[1])
            for lab, g in got.items():
                if g != want:
                    fails += 1
                    print("MISMATCH", lab, "costs", costs, "pat", pat, "k", k, "nocase", nocase, "delim", delim, "n", len(text),
                          "want", want[0], "got", g[0], "seed", seed, "case", n_cases, flush=True)
                    if fails <= 3:
                        with open(os.path.join(ROOT, "gpurun_out", "stress_fail_%d_%d.bin" % (seed, n_cases)), "wb") as f:
                            f.write(text)
    except A.AghError as e:
        print("ERROR", e, pat, k, nocase, delim, len(text), flush=True)
        fails += 1
    n_cases += 1
print("cases", n_cases, "failures", fails)
sys.exit(1 if fails else 0)
