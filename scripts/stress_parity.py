"""Randomised parity stress: random patterns / k / -i / delimiters / texts, every device engine
against the oracle.  usage: stress_parity.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import agrep_amd as A
import _oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
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
    delim = b"\n" if rng.random() < 0.8 else bytes([rng.choice(b";|\t")])
    if delim[0] in pat:
        continue
    n = rng.choice([0, 1, 5, 100, 1023, 1024, 1025, 4096, 70000, 262144, 262145, 600000])
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
    want = O.asearch(pat, k, text, delim=delim, nocase=nocase, cap=300000)
    try:
        with A.Query(pat, k, nocase=nocase, delim=delim) as q:
            got = {}
            for lab, fl, cap in (("default", 0, 300000), ("fullscan", A.FORCE_FULLSCAN, 300000),
                                 ("lean", A.COUNT, 0), ("numbered", A.COUNT | A.FORCE_NUMBERED, 0)):
                res, ms = q.scan_buffer(text, flags=fl, cap=cap)
                got[lab] = (res.n_matched, [(s, e) for s, e, _ in ms]) if cap else (res.n_matched, want[1])
            for lab, g in got.items():
                if g != want:
                    fails += 1
                    print("MISMATCH", lab, "pat", pat, "k", k, "nocase", nocase, "delim", delim, "n", len(text),
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
