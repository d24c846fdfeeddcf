"""The argument behind k_delim_bitmap's word-at-a-time form (agrep_amd/csrc/agh_sweep.hip, round 5), checked on a
model: the reference selects delimiter occurrences leftmost and non-overlapping (asearch.c:54-57, 175-186: after a
detection the delimiter positions are cleared).  The kernel computes, for 64 text bytes and the 16 in front of them,
where ANY occurrence ends (shifted ANDs of per-byte equality masks) and keeps all of those ends unless two of them lie
less than dlen apart -- only then it runs the serial selection.  Claim: an occurrence with no other occurrence ending
in the dlen - 1 positions in front of it is always selected, whatever happened further left.  This file restates both
sides in Python and compares them over random texts and delimiters that overlap themselves ("aa", "abab", "\\n\\n").
Host-only: no GPU, no library."""
import random


def greedy_ends(text, delim):
    """positions where a selected occurrence ends: scan left to right, restart after every detection"""
    ends, i, d = set(), 0, len(delim)
    while i + d <= len(text):
        if text[i:i + d] == delim:
            ends.add(i + d - 1)
            i += d
        else:
            i += 1
    return ends


def word_model(text, delim, wi):
    """what delim_bitmap_word() returns for word wi (64 positions), and whether it needed the serial path"""
    b, d, n = wi * 64, len(delim), len(text)
    lo = max(0, b - 16)
    all_ends = {e for e in range(lo + d - 1, min(n, b + 64)) if text[e - d + 1:e + 1] == delim}
    overlap = any(e >= b and any((e - p) in all_ends for p in range(1, d)) for e in all_ends)
    if overlap or wi == 0:
        return {e for e in greedy_ends(text, delim) if b <= e < b + 64}, True     # (the serial selection: exact by definition)
    return {e for e in all_ends if e >= b}, False


def test_non_overlapping_ends_are_all_selected():
    rng = random.Random(5)
    fast = serial = 0
    for it in range(400):
        sigma = rng.choice((2, 3, 4))
        alpha = bytes(rng.sample(range(97, 123), sigma))
        dlen = rng.randint(2, 6)
        delim = bytes(rng.choice(alpha) for _ in range(dlen))
        if it % 4 == 0:
            delim = (delim[:max(1, dlen // 2)] * dlen)[:dlen]          # periodic: overlaps itself
        n = rng.randint(1, 700)
        text = bytearray(rng.choice(alpha) for _ in range(n))
        for _ in range(rng.randint(0, 8)):                             # runs of the delimiter and of its first byte
            at = rng.randrange(n)
            run = delim * rng.randint(1, 4) if rng.random() < 0.6 else delim[:1] * rng.randint(2, 12)
            text[at:at + len(run)] = run
        text = bytes(text[:n])
        want = greedy_ends(text, delim)
        for wi in range((len(text) + 63) // 64):
            got, was_serial = word_model(text, delim, wi)
            assert got == {e for e in want if wi * 64 <= e < wi * 64 + 64}, (delim, wi, text)
            serial += was_serial
            fast += not was_serial
    assert fast > 500 and serial > 300          # both paths were exercised
