"""ctypes binding of include/agrep_hip.h.  Mirrors the C names one to one."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AGH_LIB_PATH: a variant build for A/B runs (e.g. make -C agrep_amd/csrc FT_BITS=14)
LIB_PATH = os.environ.get("AGH_LIB_PATH") or os.path.join(_HERE, "libagrep_hip.so")

COUNT = 0x01
FILENAMEONLY = 0x02
INVERT = 0x04
NO_BYTES = 0x200
EMIT_HEAD_DELIM = 0x400
EMIT_TAIL_DELIM = 0x800
TIME_SWEEP = 0x80
TIME_SCAN = 0x100
FORCE_FULLSCAN = 0x10
FORCE_FILTER = 0x20
FORCE_NUMBERED = 0x40
Q_NOCASE, Q_WORD, Q_WHOLELINE = 1, 2, 4
ENGINE_FULLSCAN = 1
ENGINE_FILTER = 2


class AghError(RuntimeError):
    pass


class Match(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("index", C.c_uint64)]


class PatternTables(C.Structure):
    _fields_ = [("Mask", C.c_uint32 * 256), ("Init0", C.c_uint32), ("Init1", C.c_uint32),
                ("NO_ERR_MASK", C.c_uint32), ("endposition", C.c_uint32), ("D_endpos", C.c_uint32),
                ("wildmask", C.c_uint32), ("M", C.c_int), ("AND", C.c_int), ("simple", C.c_int)]


class Result(C.Structure):
    _fields_ = [("n_matched", C.c_uint64), ("n_records", C.c_uint64), ("n_bytes", C.c_uint64),
                ("n_candidates", C.c_uint64), ("n_stored", C.c_uint64), ("engine", C.c_uint32),
                ("truncated", C.c_uint32), ("device_ms", C.c_double), ("sweep_ms", C.c_double),
                ("sweep_launches", C.c_uint32), ("lean_reruns", C.c_uint32),
                ("n_segments", C.c_uint32), ("fused_segments", C.c_uint32),
                ("copied_segments", C.c_uint32), ("reserved", C.c_uint32)]


# agh_emit_fn: int (*)(void *ctx, const agh_match *m, size_t n, const unsigned char *bytes, size_t n_bytes)
EMIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Match), C.c_size_t, C.POINTER(C.c_ubyte), C.c_size_t)

_LIB = None


def lib():
    """Load libagrep_hip.so (built by __graft_entry__.build()); never falls back."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (SONAME libamdhip64.so.7,
    # found through torch/lib's RPATH).  If torch is going to be used in this process (device
    # memory, torch.distributed) it must be loaded FIRST so that our NEEDED libamdhip64.so.7
    # resolves to the already-loaded copy; a second runtime would see no GPUs.  Plain C users
    # (agrep_amd/host) get /opt/rocm/lib through the library's RUNPATH.
    if os.environ.get("AGH_NO_TORCH_PRELOAD", "") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise AghError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u8p = C.c_void_p, C.c_char_p
    L.agh_query_literal.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
    L.agh_query_literal.restype = vp
    L.agh_query_from_maskgen.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u8p,
                                         C.c_int, C.c_int, C.c_int]
    L.agh_query_from_maskgen.restype = vp
    L.agh_query_multi.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int, u8p,
                                  C.c_int]
    L.agh_query_multi.restype = vp
    L.agh_query_literal_ex.argtypes = [u8p, C.c_int, C.c_int, C.c_uint, u8p, C.c_int]
    L.agh_query_literal_ex.restype = vp
    L.agh_query_multi_ex.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_uint, u8p, C.c_int]
    L.agh_query_multi_ex.restype = vp
    L.agh_query_multi_approx.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                         C.c_int, u8p, C.c_int]
    L.agh_query_multi_approx.restype = vp
    L.agh_compile_pattern.argtypes = [u8p, C.c_int, C.c_uint, u8p, C.c_int, C.POINTER(PatternTables)]
    L.agh_compile_pattern.restype = C.c_int
    L.agh_query_pattern.argtypes = [u8p, C.c_int, C.c_int, C.c_uint, u8p, C.c_int]
    L.agh_query_pattern.restype = vp
    L.agh_query_set_costs.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.agh_query_set_costs.restype = C.c_int
    L.agh_query_free.argtypes = [vp]
    L.agh_query_free.restype = None
    L.agh_query_info.argtypes = [vp] + [C.POINTER(C.c_int)] * 4
    L.agh_query_info.restype = C.c_int
    L.agh_scan_buffer.argtypes = [vp, vp, C.c_size_t, C.c_uint, C.POINTER(Result),
                                  C.POINTER(Match), C.c_size_t]
    L.agh_scan_buffer.restype = C.c_int
    L.agh_scan_fd.argtypes = [vp, C.c_int, C.c_uint, C.POINTER(Result), C.POINTER(Match),
                              C.c_size_t]
    L.agh_scan_fd.restype = C.c_int
    L.agh_scan_fd_emit.argtypes = [vp, C.c_int, C.c_uint, C.POINTER(Result), EMIT_FN, vp]
    L.agh_scan_fd_emit.restype = C.c_int
    L.agh_scan_fd_range_emit.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint, C.POINTER(Result), EMIT_FN, vp]
    L.agh_scan_fd_range_emit.restype = C.c_int
    L.agh_scan_device_emit.argtypes = [vp, vp, C.c_size_t, C.c_uint, C.POINTER(Result), EMIT_FN, vp]
    L.agh_scan_device_emit.restype = C.c_int
    L.agh_scan_device_reduce.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_uint, C.POINTER(Result), C.POINTER(C.c_uint64)]
    L.agh_scan_device_reduce.restype = C.c_int
    L.agh_rescan_staged.argtypes = [vp, C.c_uint, C.POINTER(Result), C.POINTER(Match), C.c_size_t]
    L.agh_rescan_staged.restype = C.c_int
    L.agh_fetch_records.argtypes = [vp, C.POINTER(Match), C.c_size_t, vp, C.c_size_t,
                                    C.POINTER(C.c_size_t)]
    L.agh_fetch_records.restype = C.c_int
    L.agh_scan_device.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint, C.POINTER(Result), vp,
                                  C.c_size_t]
    L.agh_scan_device.restype = C.c_int
    L.agh_device_count.restype = C.c_int
    L.agh_set_device.argtypes = [C.c_int]
    L.agh_set_device.restype = C.c_int
    L.agh_corpus_fill_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, vp,
                                         C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.POINTER(C.c_uint64), vp]
    L.agh_corpus_fill_device.restype = C.c_int
    L.agh_probe_read_ms.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_double)]
    L.agh_probe_read_ms.restype = C.c_int
    if hasattr(L, "agh_probe_variant_ms"):      # diagnostics build only (make -C agrep_amd/csrc EXP=1)
        L.agh_probe_variant_ms.argtypes = [vp, C.c_size_t, vp, C.c_int, C.POINTER(C.c_double)]
        L.agh_probe_variant_ms.restype = C.c_int
    L.agh_scan_fd_range.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint, C.POINTER(Result),
                                    C.POINTER(Match), C.c_size_t]
    L.agh_scan_fd_range.restype = C.c_int
    L.agh_shard_cuts_fd.argtypes = [C.c_int, u8p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.agh_shard_cuts_fd.restype = C.c_int
    L.agh_comm_unique_id.argtypes = [C.c_char_p]
    L.agh_comm_unique_id.restype = C.c_int
    L.agh_comm_init_rank.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.agh_comm_init_rank.restype = vp
    L.agh_comm_init_all.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    L.agh_comm_init_all.restype = C.c_int
    L.agh_comm_info.argtypes = [vp] + [C.POINTER(C.c_int)] * 3
    L.agh_comm_info.restype = C.c_int
    L.agh_comm_free.argtypes = [vp]
    L.agh_comm_free.restype = None
    L.agh_reduce_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.agh_reduce_counts.restype = C.c_int
    L.agh_reduce_file_hits.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.agh_reduce_file_hits.restype = C.c_int
    L.agh_last_error.restype = C.c_char_p
    L.agh_version.restype = C.c_char_p
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise AghError(lib().agh_last_error().decode("latin1"))


def device_count():
    return lib().agh_device_count()


def set_device(i):
    _check(lib().agh_set_device(i))


class Query:
    """A compiled pattern (agh_query_literal / agh_query_from_maskgen)."""

    def __init__(self, pattern, k=0, nocase=False, delim=b"\n", _handle=None, word=False, wholeline=False):
        self._h = None
        L = lib()
        if _handle is not None:
            self._h = _handle
        else:
            pattern = bytes(pattern)
            delim = bytes(delim)
            if word or wholeline:               # -w / -x guards of the simple-pattern engines
                qf = (Q_NOCASE if nocase else 0) | (Q_WORD if word else 0) | (Q_WHOLELINE if wholeline else 0)
                self._h = L.agh_query_literal_ex(pattern, len(pattern), k, qf, delim, len(delim))
            else:
                self._h = L.agh_query_literal(pattern, len(pattern), k, int(nocase), delim, len(delim))
        if not self._h:
            raise AghError(L.agh_last_error().decode("latin1"))

    @classmethod
    def from_maskgen(cls, Mask, Init0, Init1, NO_ERR_MASK, endposition, D_endpos, M, old_D_pat,
                     D, AND=0):
        arr = (C.c_uint32 * 256)(*Mask)
        old_D_pat = bytes(old_D_pat)
        h = lib().agh_query_from_maskgen(arr, Init0, Init1, NO_ERR_MASK, endposition, D_endpos,
                                         M, old_D_pat, len(old_D_pat), D, AND)
        if not h:
            raise AghError(lib().agh_last_error().decode("latin1"))
        return cls(None, _handle=h)

    @classmethod
    def pattern(cls, pattern, k=0, nocase=False, delim=b"\n", word=False, wholeline=False):
        """agh_query_pattern: agrep's non-regex pattern language compiled by the library itself"""
        qf = (Q_NOCASE if nocase else 0) | (Q_WORD if word else 0) | (Q_WHOLELINE if wholeline else 0)
        pattern, delim = bytes(pattern), bytes(delim)
        h = lib().agh_query_pattern(pattern, len(pattern), k, qf, delim, len(delim))
        if not h:
            raise AghError(lib().agh_last_error().decode("latin1"))
        return cls(None, _handle=h)

    @classmethod
    def multi(cls, patterns, nocase=False, delim=b"\n", k=0, word=False, wholeline=False):
        """-f: multi-pattern query; exact (agh_query_multi) or with k errors
        (agh_query_multi_approx)."""
        pats = [bytes(p) for p in patterns]
        arr = (C.c_char_p * len(pats))(*pats)
        lens = (C.c_int * len(pats))(*[len(p) for p in pats])
        if k:
            h = lib().agh_query_multi_approx(arr, lens, len(pats), int(k), int(nocase), bytes(delim),
                                             len(delim))
        elif word or wholeline:
            qf = (Q_NOCASE if nocase else 0) | (Q_WORD if word else 0) | (Q_WHOLELINE if wholeline else 0)
            h = lib().agh_query_multi_ex(arr, lens, len(pats), qf, bytes(delim), len(delim))
        else:
            h = lib().agh_query_multi(arr, lens, len(pats), int(nocase), bytes(delim), len(delim))
        if not h:
            raise AghError(lib().agh_last_error().decode("latin1"))
        return cls(None, _handle=h)

    def set_costs(self, insertion=1, substitution=1, deletion=1):
        _check(lib().agh_query_set_costs(self._h, insertion, substitution, deletion))
        return self

    def info(self):
        m, d, fq, fh = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(lib().agh_query_info(self._h, C.byref(m), C.byref(d), C.byref(fq), C.byref(fh)))
        return {"m": m.value, "k": d.value, "filter_q": fq.value, "filter_h": fh.value}

    def scan_buffer(self, text, flags=0, cap=0):
        """text: bytes or a contiguous numpy uint8 array -> (Result, [(start, end, index)])"""
        import numpy as np
        if isinstance(text, np.ndarray):
            a = np.ascontiguousarray(text, dtype=np.uint8)
            ptr, n, keep = a.ctypes.data, a.size, a
        else:
            b = bytes(text)
            keep = C.create_string_buffer(b, len(b))
            ptr, n = C.addressof(keep), len(b)
        res = Result()
        ms = (Match * max(cap, 1))()
        _check(lib().agh_scan_buffer(self._h, ptr, n, flags, C.byref(res), ms if cap else None,
                                     cap))
        return res, [(ms[i].start, ms[i].end, ms[i].index) for i in range(int(res.n_stored))]

    def scan_fd(self, fd, flags=0, cap=0):
        res = Result()
        ms = (Match * max(cap, 1))()
        _check(lib().agh_scan_fd(self._h, fd, flags, C.byref(res), ms if cap else None, cap))
        return res, [(ms[i].start, ms[i].end, ms[i].index) for i in range(int(res.n_stored))]

    def scan_fd_range(self, fd, begin, end, flags=0, cap=0):
        """One rank's shard of a seekable file: bytes [begin, end)."""
        res = Result()
        ms = (Match * max(cap, 1))()
        _check(lib().agh_scan_fd_range(self._h, fd, begin, end, flags, C.byref(res),
                                       ms if cap else None, cap))
        return res, [(ms[i].start, ms[i].end, ms[i].index) for i in range(int(res.n_stored))]

    def _emit_collector(self, on_batch, stop_after, summarize=False, hasher=None):
        """-> (callback object, list of batches); a batch = ([(start, end, index)], [record bytes] or None),
        or with summarize (timing runs: no Python object per record) (records, bytes, first start, last end);
        hasher (with summarize): a hashlib object fed with the raw bytes of every emit() call"""
        batches = []

        def cb(ctx, m, n, bytes_, n_bytes):
            if summarize:
                batches.append((n, n_bytes, m[0].start if n else 0, m[n - 1].end if n else 0))
                if hasher is not None and n_bytes:
                    hasher.update(C.string_at(bytes_, n_bytes))
                return 0
            ms = [(m[i].start, m[i].end, m[i].index) for i in range(n)]
            recs = None
            if bytes_:
                raw = C.string_at(bytes_, n_bytes)
                recs, o = [], 0
                for s_, e_, _ in ms:
                    recs.append(raw[o:o + (e_ - s_)])
                    o += e_ - s_
            batches.append((ms, recs))
            if on_batch:
                on_batch(ms, recs)
            return 1 if (stop_after is not None and len(batches) >= stop_after) else 0
        return EMIT_FN(cb), batches

    def scan_fd_emit(self, fd, flags=0, on_batch=None, stop_after=None, byte_range=None, summarize=False, hasher=None):
        """agh_scan_fd_emit / agh_scan_fd_range_emit -> (Result, batches): record output while the input
        streams through bounded device segments; one batch per segment, in file order."""
        cb, batches = self._emit_collector(on_batch, stop_after, summarize, hasher)
        res = Result()
        if byte_range is None:
            _check(lib().agh_scan_fd_emit(self._h, fd, flags, C.byref(res), cb, None))
        else:
            _check(lib().agh_scan_fd_range_emit(self._h, fd, byte_range[0], byte_range[1], flags, C.byref(res), cb, None))
        return res, batches

    def scan_device_emit(self, dev_ptr, n, flags=0, on_batch=None, summarize=False, hasher=None):
        """agh_scan_device_emit: matched records (offsets, numbers, bytes) of text resident in HBM"""
        cb, batches = self._emit_collector(on_batch, None, summarize, hasher)
        res = Result()
        _check(lib().agh_scan_device_emit(self._h, dev_ptr, n, flags, C.byref(res), cb, None))
        return res, batches

    def fetch_records(self, matches):
        """matches: [(start, end, index)] from the last scan_fd / scan_buffer -> list of bytes"""
        n = len(matches)
        ms = (Match * max(n, 1))()
        for i, (s, e, idx) in enumerate(matches):
            ms[i].start, ms[i].end, ms[i].index = s, e, idx
        total = sum(e - s for s, e, _ in matches)
        buf = C.create_string_buffer(max(total, 1))
        got = C.c_size_t()
        _check(lib().agh_fetch_records(self._h, ms, n, C.addressof(buf), total, C.byref(got)))
        raw, out, o = buf.raw[:total], [], 0
        for s, e, _ in matches:
            out.append(raw[o:o + (e - s)])
            o += e - s
        return out

    def scan_device(self, dev_ptr, n, stream=None, flags=0, match_pos_ptr=None, match_cap=0,
                    time_sweep=True, time_scan=True):
        """dev_ptr: device address (e.g. torch tensor .data_ptr()), n bytes.  time_sweep: ask
        for Result.sweep_ms / device_ms (AGH_TIME_SWEEP, AGH_TIME_SCAN: two HIP events per scan
        each) -- on by default in
        this test / measurement binding, off by default in the C-ABI."""
        if time_sweep:
            flags |= TIME_SWEEP
        if time_scan:
            flags |= TIME_SCAN
        res = Result()
        _check(lib().agh_scan_device(self._h, dev_ptr, n, stream, flags, C.byref(res),
                                     match_pos_ptr, match_cap))
        return res

    def scan_device_reduce(self, comm, dev_ptr, n, flags=COUNT, stream=None, time_sweep=False):
        """agh_scan_device_reduce: this rank's count-only scan + the sum of (matched, records) over all ranks
        of the communicator, all-reduced on the scan's stream (one host sync) -> (Result, (matched, records))"""
        if time_sweep:
            flags |= TIME_SWEEP
        res = Result()
        tot = (C.c_uint64 * 2)()
        _check(lib().agh_scan_device_reduce(self._h, comm._h, dev_ptr, n, stream, flags, C.byref(res), tot))
        return res, (int(tot[0]), int(tot[1]))

    def close(self):
        if self._h:
            lib().agh_query_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def compile_pattern(pattern, nocase=False, delim=b"\n", word=False, wholeline=False):
    """agh_compile_pattern (host-only) -> PatternTables in maskgen's layout"""
    qf = (Q_NOCASE if nocase else 0) | (Q_WORD if word else 0) | (Q_WHOLELINE if wholeline else 0)
    pattern, delim = bytes(pattern), bytes(delim)
    t = PatternTables()
    _check(lib().agh_compile_pattern(pattern, len(pattern), qf, delim, len(delim), C.byref(t)))
    return t


def shard_cuts_fd(fd, nranks, delim=b"\n"):
    """Record-aligned shard boundaries of a file (agh_shard_cuts_fd; host-only code)."""
    cuts = (C.c_uint64 * (nranks + 1))()
    _check(lib().agh_shard_cuts_fd(fd, bytes(delim), len(delim), nranks, cuts))
    return list(cuts)


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)


class Comm:
    """RCCL communicator of the C-ABI (one process per GPU): agh_comm_init_rank."""

    def __init__(self, unique_id, nranks, rank, _handle=None, _keep=None):
        self._keep = _keep
        self._h = _handle if _handle is not None else lib().agh_comm_init_rank(bytes(unique_id), nranks, rank)
        if not self._h:
            raise AghError(lib().agh_last_error().decode("latin1"))

    @classmethod
    def custom(cls, allreduce, nranks, rank):
        """agh_comm_init_custom: allreduce(values: list[int], elem_bytes) -> list[int] over the caller's transport"""
        def cb(ctx, buf, count, elem):
            try:
                arr = (C.c_uint64 * count).from_address(buf) if elem == 8 else (C.c_uint8 * count).from_address(buf)
                out = allreduce([int(x) for x in arr], elem)
                for i, v in enumerate(out):
                    arr[i] = int(v)
                return 0
            except Exception:                               # a Python exception must not cross the C frames
                return 1
        fn = ALLREDUCE_FN(cb)
        L = lib()
        L.agh_comm_init_custom.argtypes = [ALLREDUCE_FN, C.c_void_p, C.c_int, C.c_int]
        L.agh_comm_init_custom.restype = C.c_void_p
        return cls(None, nranks, rank, _handle=L.agh_comm_init_custom(fn, None, nranks, rank), _keep=fn)

    def info(self):
        r, n, d = C.c_int(), C.c_int(), C.c_int()
        _check(lib().agh_comm_info(self._h, C.byref(r), C.byref(n), C.byref(d)))
        return {"rank": r.value, "nranks": n.value, "device": d.value}

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().agh_comm_unique_id(buf))
        return buf.raw

    def reduce_counts(self, n_matched, n_records=0):
        v = (C.c_uint64 * 2)(int(n_matched), int(n_records))
        _check(lib().agh_reduce_counts(self._h, v))
        return int(v[0]), int(v[1])

    def reduce_file_hits(self, hits):
        buf = C.create_string_buffer(bytes(1 if h else 0 for h in hits), len(hits))
        _check(lib().agh_reduce_file_hits(self._h, buf, len(hits)))
        return [b != 0 for b in buf.raw[:len(hits)]]

    def close(self):
        if self._h:
            lib().agh_comm_free(self._h)
            self._h = None


def corpus_fill_device(dev_ptr, n_pages, first_page=0, seed=12345, variants=(), plant_period=500,
                       upper_permille=0, stream=None):
    """Fill n_pages*4096 bytes at dev_ptr with the synthetic corpus; -> planted counts."""
    nv = len(variants)
    vbuf = (C.c_uint8 * (80 * max(nv, 1)))()
    vlen = (C.c_uint32 * 8)()
    for i, v in enumerate(variants):
        vlen[i] = len(v)
        for j, ch in enumerate(v):
            vbuf[i * 80 + j] = ch
    planted = (C.c_uint64 * 8)()
    _check(lib().agh_corpus_fill_device(dev_ptr, first_page, n_pages, seed, C.addressof(vbuf),
                                        vlen, nv, plant_period, upper_permille, planted, stream))
    return list(planted)[:max(nv, 1)]


def probe_read_ms(dev_ptr, n, stream=None):
    ms = C.c_double()
    _check(lib().agh_probe_read_ms(dev_ptr, n, stream, C.byref(ms)))
    return ms.value


def probe_variant_ms(dev_ptr, n, exp, stream=None):
    ms = C.c_double()
    _check(lib().agh_probe_variant_ms(dev_ptr, n, stream, exp, C.byref(ms)))
    return ms.value
