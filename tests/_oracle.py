"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class OrcTables(C.Structure):
    _fields_ = [("Mask", C.c_uint32 * 256), ("Init0", C.c_uint32), ("Init1", C.c_uint32),
                ("NO_ERR_MASK", C.c_uint32), ("endposition", C.c_uint32),
                ("D_endpos", C.c_uint32), ("wildmask", C.c_uint32), ("M", C.c_int32),
                ("D_length", C.c_int32), ("AND", C.c_int32)]


class OrcRecord(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64)]


class CgParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_variants", C.c_uint32), ("plant_period", C.c_uint32),
                ("upper_permille", C.c_uint32), ("vlen", C.c_uint32 * 8),
                ("variants", (C.c_uint8 * 80) * 8)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        L = C.CDLL(path)
        u8p = C.c_char_p
        L.orc_maskgen_literal.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.POINTER(OrcTables)]
        L.orc_maskgen_literal.restype = C.c_int
        L.orc_asearch.argtypes = [C.POINTER(OrcTables), C.c_int, C.c_void_p, C.c_size_t, u8p,
                                  C.c_int, C.POINTER(OrcRecord), C.c_size_t]
        L.orc_asearch.restype = C.c_int64
        L.orc_asearch_costs.argtypes = [C.POINTER(OrcTables), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_size_t, u8p, C.c_int,
                                        C.POINTER(OrcRecord), C.c_size_t]
        L.orc_asearch_costs.restype = C.c_int64
        L.orc_sgrep_verify.argtypes = [u8p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int,
                                       C.POINTER(OrcRecord), C.c_size_t]
        L.orc_sgrep_verify.restype = C.c_int64
        L.orc_dp_count.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, u8p,
                                   C.c_int, C.POINTER(OrcRecord), C.c_size_t]
        L.orc_dp_count.restype = C.c_int64
        L.orc_dp_best.argtypes = [u8p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_dp_best.restype = C.c_int
        L.orc_wm_count.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_size_t, u8p, C.c_int, C.POINTER(OrcRecord), C.c_size_t]
        L.orc_wm_count.restype = C.c_int64
        L.orc_multi_exact_count.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int,
                                            C.c_int, C.c_void_p, C.c_size_t, u8p, C.c_int,
                                            C.POINTER(OrcRecord), C.c_size_t]
        L.orc_multi_exact_count.restype = C.c_int64
        L.cg_fill.argtypes = [C.POINTER(CgParams), C.c_uint64, C.c_uint64, C.c_void_p,
                              C.POINTER(C.c_uint64)]
        L.cg_fill.restype = None
        _LIB = L
    return _LIB


def _buf(text):
    """bytes / numpy uint8 -> (pointer, length, keepalive)"""
    if isinstance(text, np.ndarray):
        a = np.ascontiguousarray(text, dtype=np.uint8)
        return a.ctypes.data, a.size, a
    b = bytes(text)
    keep = C.create_string_buffer(b, len(b))
    return C.addressof(keep), len(b), keep


def _recs(cap):
    return (OrcRecord * max(cap, 1))()


def _collect(recs, n, cap):
    return [(recs[i].start, recs[i].end) for i in range(min(n, cap))]


def maskgen(pat, delim=b"\n", nocase=False):
    t = OrcTables()
    M = lib().orc_maskgen_literal(pat, len(pat), delim, len(delim), int(nocase), C.byref(t))
    return M, t


def asearch(pat, k, text, delim=b"\n", nocase=False, cap=0):
    M, t = maskgen(pat, delim, nocase)
    if M < 0:
        raise ValueError("pattern too long for the reference word")
    p, n, keep = _buf(text)
    recs = _recs(cap)
    cnt = lib().orc_asearch(C.byref(t), k, p, n, delim, len(delim), recs, cap)
    return cnt, _collect(recs, cnt, cap)


def tables_from_golden(g, M, dlen=1):
    """OrcTables from a tests/golden 'tables' object (the reference's own maskgen output)."""
    t = OrcTables()
    for i, v in enumerate(g["Mask"]):
        t.Mask[i] = v
    t.Init0, t.Init1, t.NO_ERR_MASK = g["Init0"], g["Init1"], g["NO_ERR_MASK"]
    t.endposition, t.D_endpos, t.wildmask = g["endposition"], g["D_endpos"], g["wildmask"]
    t.M, t.D_length, t.AND = M, dlen, g["AND"]
    return t


def asearch_tables(t, k, text, delim=b"\n", cap=0):
    p, n, keep = _buf(text)
    recs = _recs(cap)
    cnt = lib().orc_asearch(C.byref(t), k, p, n, delim, len(delim), recs, cap)
    return cnt, _collect(recs, cnt, cap)


def asearch_tables_costs(t, k, costs, text, delim=b"\n", cap=0):
    """asearch1.c's recurrence on given maskgen tables (wildcards / AND / OR together with -I -S -D)"""
    p, n, keep = _buf(text)
    recs = _recs(cap)
    cnt = lib().orc_asearch_costs(C.byref(t), k, costs[0], costs[1], costs[2], p, n, delim,
                                  len(delim), recs, cap)
    return cnt, _collect(recs, cnt, cap)


def asearch_costs(pat, k, costs, text, delim=b"\n", nocase=False, cap=0):
    """costs = (I, S, D): insertion, substitution, deletion (asearch1.c)"""
    M, t = maskgen(pat, delim, nocase)
    if M < 0:
        raise ValueError("pattern too long for the reference word")
    p, n, keep = _buf(text)
    recs = _recs(cap)
    cnt = lib().orc_asearch_costs(C.byref(t), k, costs[0], costs[1], costs[2], p, n, delim,
                                  len(delim), recs, cap)
    return cnt, _collect(recs, cnt, cap)


def sgrep_verify(pat, k, text, clean=False, cap=0):
    p, n, keep = _buf(text)
    recs = _recs(cap)
    cnt = lib().orc_sgrep_verify(pat, len(pat), k, p, n, int(clean), recs, cap)
    return cnt, _collect(recs, cnt, cap)


def dp_count(pat, k, text, delim=b"\n", nocase=False, cap=0):
    p, n, keep = _buf(text)
    recs = _recs(cap)
    cnt = lib().orc_dp_count(pat, len(pat), k, int(nocase), p, n, delim, len(delim), recs, cap)
    return cnt, _collect(recs, cnt, cap)


def dp_best(pat, rec, nocase=False):
    p, n, keep = _buf(rec)
    return lib().orc_dp_best(pat, len(pat), int(nocase), p, n)


def wm_count(pat, k, text, delim=b"\n", nocase=False, word_bits=64, cap=0):
    p, n, keep = _buf(text)
    recs = _recs(cap)
    cnt = lib().orc_wm_count(pat, len(pat), k, int(nocase), word_bits, p, n, delim, len(delim),
                             recs, cap)
    return cnt, _collect(recs, cnt, cap)


def multi_exact_count(pats, text, delim=b"\n", nocase=False, cap=0):
    p, n, keep = _buf(text)
    arr = (C.c_char_p * len(pats))(*pats)
    lens = (C.c_int * len(pats))(*[len(x) for x in pats])
    recs = _recs(cap)
    cnt = lib().orc_multi_exact_count(arr, lens, len(pats), int(nocase), p, n, delim, len(delim),
                                      recs, cap)
    return cnt, _collect(recs, cnt, cap)


def corpus_params(seed=12345, variants=(), plant_period=500, upper_permille=0):
    p = CgParams()
    p.seed = seed
    p.n_variants = len(variants)
    p.plant_period = plant_period
    p.upper_permille = upper_permille
    for i, v in enumerate(variants):
        assert len(v) <= 80
        p.vlen[i] = len(v)
        for j, ch in enumerate(v):
            p.variants[i][j] = ch
    return p


def corpus(n_pages, first_page=0, **kw):
    """-> (numpy uint8 array of n_pages*4096 bytes, planted-per-variant list)"""
    p = corpus_params(**kw)
    out = np.empty(n_pages * 4096, dtype=np.uint8)
    planted = (C.c_uint64 * 8)()
    lib().cg_fill(C.byref(p), first_page, n_pages, out.ctypes.data, planted)
    return out, list(planted)[: max(p.n_variants, 1)]


# the headline pattern and its seven planted variants ({0,0,1,1,2,2,2} edits)
PATTERN_C2 = b"approximatematch"
VARIANTS_C2 = (b"approximatematch", b"approximatematch", b"aproximatematch", b"approxXmatematch",
               b"approximatemmatcZ", b"apprximatemtch", b"appQoximRtematch")

REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "agrep")) and \
        os.path.exists(os.path.join(REF_DIR, "ref_harness"))
