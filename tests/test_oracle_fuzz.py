"""Differential fuzzing of the oracle's independent restatements against each other and,
when oracle/_ref exists (this container), against the live reference binary."""
import os
import random
import subprocess
import tempfile

import pytest

import _oracle as O


from _cases import _rand_case  # noqa: E402


@pytest.mark.parametrize("sigma", [2, 4, 27])
def test_restatements_agree_with_dp(sigma):
    rng = random.Random(1000 + sigma)
    for it in range(120):
        pat, k, text = _rand_case(rng, sigma)
        n_dp, r_dp = O.dp_count(pat, k, text, cap=1000)
        assert O.asearch(pat, k, text, cap=1000) == (n_dp, r_dp), (pat, k, text)
        for wb in (8, 16, 32, 64):
            # narrow words force the multi-word carry path for every m > 8
            assert O.wm_count(pat, k, text, word_bits=wb, cap=1000) == (n_dp, r_dp), (pat, k, wb)
        if len(pat) <= 23 and k > 0:
            assert O.sgrep_verify(pat, k, text, clean=True, cap=1000) == (n_dp, r_dp)


def test_multiword_long_patterns_agree_with_dp():
    """m up to 64 (config C3 shape: m=48, k=3, -i): no reference, DP is the anchor."""
    rng = random.Random(48)
    for it in range(60):
        pat, k, text = _rand_case(rng, 27, max_m=64)
        k = min(k, 4)
        text_u = bytes(c - 32 if (97 <= c <= 122 and rng.random() < 0.5) else c for c in text)
        want = O.dp_count(pat, k, text_u, nocase=True, cap=1000)
        for wb in (8, 32, 64):
            assert O.wm_count(pat, k, text_u, nocase=True, word_bits=wb, cap=1000) == want


def test_custom_delimiters_agree_with_dp():
    rng = random.Random(5)
    # delimiters disjoint from the pattern alphabet: see test_q11_... for why
    for delim in (b"FROM ", b"\n\n", b"#%", b";", b"@@@"):
        for it in range(40):
            pat, k, text = _rand_case(rng, 6, max_m=20)
            text = text.replace(b"\n", delim if rng.random() < 0.7 else b"\n")
            n_dp, r_dp = O.dp_count(pat, k, text, delim=delim, cap=1000)
            assert O.asearch(pat, k, text, delim=delim, cap=1000) == (n_dp, r_dp), (delim, pat, k, text)


def test_q11_delimiter_bytes_are_text_for_the_asearch_automaton():
    """Classified deviation (Q11, found here): asearch.c resets only AFTER the last delimiter
    byte and re-feeds it (asearch.c:175-186), so the bytes of a multi-byte delimiter are
    ordinary text for the error levels.  With a delimiter that shares letters with the
    pattern and k close to m the reference's asearch path can report a record whose own
    text does not match.  The product implements the record-text-only semantic (DP)."""
    pat, k, delim = b"avvvuuva", 7, b"ab"
    text = b"zzzzzzzzzzzzzzzz" + delim          # one record of z's, then an empty tail
    assert O.dp_count(pat, k, text, delim=delim)[0] == 0
    assert O.asearch(pat, k, text, delim=delim)[0] >= 1


@pytest.mark.ref
@pytest.mark.skipif(not O.have_ref(), reason="compiled reference not present (oracle/_ref)")
@pytest.mark.parametrize("sigma", [4, 27])
def test_asearch_path_of_live_reference_equals_dp(sigma):
    """-i forces the reference onto asearch.c (checksg.c:129); memory mode is free of Q1."""
    rng = random.Random(77 + sigma)
    harness = os.path.join(O.REF_DIR, "ref_harness")
    for it in range(40):
        pat, k, text = _rand_case(rng, sigma)
        if k == 0:
            k = 1
        if not text.endswith(b"\n"):
            text += b"\n"
        with tempfile.NamedTemporaryFile(delete=False) as tf:
            tf.write(text)
        try:
            out = subprocess.run([harness, "count", tf.name, "-%d" % k, "-i", pat.decode()],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout
        finally:
            os.unlink(tf.name)
        assert int(out.split()[0]) == O.dp_count(pat, k, text, nocase=True)[0], (pat, k, text)


def test_multi_pattern_with_errors_ground_truth_is_the_dp_union():
    """What the GPU tests of agh_query_multi_approx compare against (the union over the patterns
    of the single-pattern automaton) equals the definition: a record matches iff some substring
    is within edit distance k of some pattern (Sellers DP), on newline-delimited text."""
    rng = random.Random(2024)
    for it in range(40):
        sigma = rng.choice(["ab", "abcd", "abcdefghijklmnop"])
        k = rng.randint(0, 2)
        pats = sorted({"".join(rng.choice(sigma) for _ in range(rng.randint(k + 1, 9))).encode()
                       for _ in range(rng.randint(1, 6))})
        text = "".join(rng.choice(sigma + "\n") for _ in range(rng.randint(0, 3000))).encode()
        a, d = set(), set()
        for p in pats:
            a.update(O.asearch(p, k, text, cap=10000)[1])
            d.update(O.dp_count(p, k, text, cap=10000)[1])
        assert a == d, (pats, k, text[:200])
        if k == 0:
            assert sorted(a) == O.multi_exact_count(pats, text, cap=10000)[1]
