"""Record output on the device (agrep_amd/csrc/agh_records.hip): the list of matched records comes out of an
ordered compaction of the record bitmap -- file order without a sort -- with bounds, bytes and, on request, the
delimiters around every record (the buffer shape asearch.c:162-170 hands to output()); the reference's own front
end prints from that stream while the input is still being read (agrep_amd/host/ref_shim.c)."""
import os
import select
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(O.REF_DIR, "agrep")
GPU = os.path.join(O.REF_DIR, "agrep_gpu")


@pytest.fixture(scope="module")
def agh():
    import agrep_amd
    assert agrep_amd.device_count() >= 1, "GPU tests need a HIP device"
    return agrep_amd


def _raw_collector():
    got = {"ms": [], "raw": []}

    def on_batch(ms, recs):
        got["ms"].extend(ms)
    return got, on_batch


def _emit_raw(agh, q, fd, flags):
    """-> (Result, [(start, end, index)], the bytes of all emit() calls joined)"""
    import ctypes as C
    from agrep_amd import _ffi
    ms, raw = [], []

    def cb(ctx, m, n, bytes_, n_bytes):
        ms.extend((m[i].start, m[i].end, m[i].index) for i in range(n))
        raw.append(C.string_at(bytes_, n_bytes) if n_bytes else b"")
        return 0
    res = _ffi.Result()
    _ffi._check(_ffi.lib().agh_scan_fd_emit(q._h, fd, flags, C.byref(res), _ffi.EMIT_FN(cb), None))
    return res, ms, b"".join(raw)


@pytest.mark.parametrize("seg_mb", ["1024", "1"])
@pytest.mark.parametrize("delim", [b"\n", b"\r\n", b"From "])
def test_emit_with_the_delimiters_around_every_record(agh, tmp_path, monkeypatch, seg_mb, delim):
    """AGH_EMIT_TAIL_DELIM: record + the dlen bytes behind it (the appended delimiter behind a last, unterminated
    record: asearch.c:87-91); AGH_EMIT_HEAD_DELIM: min(dlen, start) bytes in front of it -- also for the first
    record of a later stream segment, whose delimiter lies in the segment before (1 MiB segments here)."""
    monkeypatch.setenv("AGH_STREAM_SEG_MB", seg_mb)
    text, _ = O.corpus((6 << 20) // 4096, seed=77, variants=O.VARIANTS_C2, plant_period=40)
    tb = text.tobytes()
    if delim != b"\n":
        tb = tb.replace(b"\n", delim)
    for body in (tb, tb[:-len(delim)], tb[:-len(delim)] + delim[:-1] if len(delim) > 1 else tb[:-1]):
        want_n, want = O.asearch(O.PATTERN_C2, 2, body, cap=400000, delim=delim)
        p = tmp_path / "t.txt"
        p.write_bytes(body)
        with agh.Query(O.PATTERN_C2, 2, delim=delim) as q:
            out = {}
            for name, fl in (("plain", 0), ("tail", agh.EMIT_TAIL_DELIM), ("both", agh.EMIT_HEAD_DELIM | agh.EMIT_TAIL_DELIM)):
                fd = os.open(str(p), os.O_RDONLY)
                try:
                    out[name] = _emit_raw(agh, q, fd, fl)
                finally:
                    os.close(fd)
        stream = body + delim                           # what the engines see: the delimiter appended at the end
        dl = len(delim)
        for name in out:
            res, ms, raw = out[name]
            assert res.n_matched == want_n and [(s, e) for s, e, _ in ms] == want, (name, seg_mb, delim)
        assert out["plain"][2] == b"".join(body[s:e] for s, e in want)
        assert out["tail"][2] == b"".join(stream[s:e + dl] for s, e in want)
        assert out["both"][2] == b"".join(stream[max(s - dl, 0):e + dl] for s, e in want)


def test_device_list_is_in_file_order(agh):
    """agh_scan_device with a position array: one offset inside every matched record, ascending (the ordered
    compaction of the record bitmap), for the filter engine, the full scan, -f and -v"""
    import torch
    text, _ = O.corpus((8 << 20) // 4096, seed=5, variants=O.VARIANTS_C2, plant_period=25)
    tb = text.tobytes()
    dev = torch.from_numpy(text).cuda()
    pos = torch.zeros(200000, dtype=torch.int64, device="cuda")
    nl = np.flatnonzero(text == 10)
    want_n, want = O.asearch(O.PATTERN_C2, 2, tb, cap=200000)
    starts = np.array([s for s, _ in want])
    for fl in (0, agh.FORCE_FULLSCAN):
        with agh.Query(O.PATTERN_C2, 2) as q:
            res = q.scan_device(dev.data_ptr(), len(tb), flags=fl, match_pos_ptr=pos.data_ptr(), match_cap=200000)
        assert res.n_matched == want_n == res.n_stored
        got = pos[:want_n].cpu().numpy()
        assert (np.diff(got) > 0).all()
        # the record around every position is the i-th matched record
        rec_start = np.where(np.searchsorted(nl, got) > 0, nl[np.maximum(np.searchsorted(nl, got) - 1, 0)] + 1, 0)
        assert (rec_start == starts).all()
    # a capacity below the number of matches: the FIRST records in file order, truncated reported
    with agh.Query(O.PATTERN_C2, 2) as q:
        res, ms = q.scan_buffer(text, cap=1000)
    assert res.truncated and res.n_stored == 1000 and res.n_matched == want_n
    assert [(s, e) for s, e, _ in ms] == want[:1000]


def test_inverse_list_above_one_emit_piece(agh, tmp_path):
    """-v over 1.3 M records: more than one emit() piece (2^20 records each), file order kept, every record once"""
    text, _ = O.corpus((104 << 20) // 4096, seed=3, variants=O.VARIANTS_C2, plant_period=500)
    tb = text.tobytes()
    want_n, want = O.asearch(O.PATTERN_C2, 2, tb, cap=100000)
    matched_starts = set(s for s, _ in want)
    nl = np.flatnonzero(text == 10)
    p = tmp_path / "v.txt"
    p.write_bytes(tb)
    calls = []
    with agh.Query(O.PATTERN_C2, 2) as q:
        fd = os.open(str(p), os.O_RDONLY)
        try:
            res, batches = q.scan_fd_emit(fd, flags=agh.INVERT | agh.EMIT_TAIL_DELIM, summarize=True)
        finally:
            os.close(fd)
    assert len(batches) >= 2 and len(nl) > (1 << 20)
    assert sum(b[0] for b in batches) == len(nl) - want_n == res.n_matched
    assert sum(b[1] for b in batches) == len(tb) - sum(e - s + 1 for s, e in want)
    firsts = [b[2] for b in batches]
    assert firsts == sorted(firsts) and firsts[0] == (0 if 0 not in matched_starts else firsts[0])


def test_inverse_list_of_a_fresh_query_over_short_lines(agh):
    """-v with a record list on a query that has not scanned before, over 8-byte lines: the first attempt sizes the
    record bitmap (and rec_pos, one entry per bit) from a hint of one record per 64 bytes, the text has eight times
    as many -- the queued compaction must stay inside what was allocated (it clamps the record count to the bitmap's
    bits) and the rerun with the right size lists every non-matching record once, in file order."""
    n_lines = 1 << 20
    rows = np.full((n_lines, 8), ord("x"), dtype=np.uint8)
    rows[:, 7] = 10
    hit = np.arange(5, n_lines, 997)
    rows[hit, :7] = np.frombuffer(b"matchme", dtype=np.uint8)
    text = rows.reshape(-1)
    for delim_flags in (0, agh.EMIT_TAIL_DELIM):
        with agh.Query(b"matchme", 0) as q:                       # fresh: no bitmap hint from an earlier scan
            import torch
            dev = torch.from_numpy(text).cuda()
            res, batches = q.scan_device_emit(dev.data_ptr(), text.size, flags=agh.INVERT | delim_flags, summarize=True)
            assert res.n_matched == n_lines - hit.size
            assert sum(b[0] for b in batches) == n_lines - hit.size
            assert sum(b[1] for b in batches) == (n_lines - hit.size) * (8 if delim_flags else 7)
            # ... and the plain list afterwards (the hint is right now), then -v again on the grown buffers
            res2, b2 = q.scan_device_emit(dev.data_ptr(), text.size, flags=delim_flags, summarize=True)
            assert res2.n_matched == hit.size == sum(b[0] for b in b2)
            res3, b3 = q.scan_device_emit(dev.data_ptr(), text.size, flags=agh.INVERT | delim_flags, summarize=True)
            assert sum(b[0] for b in b3) == n_lines - hit.size
            # starts ascend over the pieces
            firsts = [b[2] for b in batches]
            assert firsts == sorted(firsts)
    with agh.Query(b"matchme", 0) as q:
        res, ms = q.scan_buffer(text[: 8 * 4096], flags=agh.INVERT, cap=8192)
        want = [i * 8 for i in range(4096) if i not in set(hit.tolist())]
        assert [m[0] for m in ms] == want


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(GPU)), reason="oracle/_ref/agrep and agrep_gpu not built")
def test_shim_prints_while_the_input_is_still_being_read(tmp_path):
    """The reference's front end on the GPU engines (agrep_gpu) prints the matched records of the first stream
    segments before the input has ended (asearch.c:162-170 calls output() from inside its block loop): the writer
    of a pipe holds back the last third until records have arrived.  Byte-identical to the all-CPU reference."""
    text, _ = O.corpus((96 << 20) // 4096, seed=41, variants=O.VARIANTS_C2, plant_period=20)
    tb = text.tobytes()
    p = tmp_path / "in.txt"
    p.write_bytes(tb)
    ref = subprocess.run([REF, "-V0", "-2", O.PATTERN_C2.decode(), str(p)], stdout=subprocess.PIPE).stdout
    env = dict(os.environ, AGH_STREAM_SEG_MB="16")
    for args in (["-V0", "-2"], ["-V0", "-2", "-n"]):
        if "-n" in args:
            ref = subprocess.run([REF] + args + [O.PATTERN_C2.decode(), str(p)], stdout=subprocess.PIPE).stdout
        # (the reference's front end wants a file argument: /dev/stdin)
        pr = subprocess.Popen([GPU] + args + [O.PATTERN_C2.decode(), "/dev/stdin"], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, env=env)
        got = []
        early = {"bytes": 0}
        hold = (len(tb) * 2 // 3) & ~4095

        def feed():
            try:
                pr.stdin.write(tb[:hold])
                pr.stdin.flush()
                t0 = time.time()
                while early["bytes"] == 0 and time.time() - t0 < 60:      # wait for the first records
                    time.sleep(0.01)
                early["seen_before_the_rest"] = early["bytes"]
                pr.stdin.write(tb[hold:])
                pr.stdin.close()
            except BrokenPipeError:
                pass
        th = threading.Thread(target=feed)
        th.start()
        while True:
            chunk = pr.stdout.read1(1 << 20)
            if not chunk:
                break
            got.append(chunk)
            early["bytes"] += len(chunk)
        th.join()
        pr.wait()
        assert early.get("seen_before_the_rest", 0) > 0, "no record came out before the input ended"
        assert b"".join(got) == ref, args
    # a file (pread readers, 16 MiB segments) gives the same bytes
    out = subprocess.run([GPU, "-V0", "-2", "-n", O.PATTERN_C2.decode(), str(p)], stdout=subprocess.PIPE, env=env).stdout
    assert out == ref


def test_shipped_defaults_in_a_fresh_process(agh):
    """tests/conftest.py forces AGH_FUSED_MIN_MB=0, AGH_TF_FAST_MIN_MB=0 and AGH_ENV_LIVE=1 for the suite (A/B
    switches: every engine form at every size).  This test runs a seeded part of the parity cases and a >= 4 GiB
    fused count in a process WITHOUT any AGH_* variable: the switches as users get them."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("AGH_")}
    env["AGH_REQUIRE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shipped_defaults_check.py")], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = r.stdout.decode(errors="replace")[-2000:]
    assert r.returncode == 0, tail
    assert "shipped defaults ok" in tail, tail


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/agrep not built")
@pytest.mark.parametrize("pattern,k", [(b"car", 0), (b"car", 1), (b"well-known", 0), (b"well-known", 1), (b"we\\ll", 1)])
def test_word_guard_next_to_high_bytes_follows_the_reference(agh, tmp_path, pattern, k):
    """-w beside bytes 0x80..0xFF (UTF-8, Latin-1): with errors, an escape or a '-' the reference goes through
    maskgen(), whose word-boundary class (maskgen.c:176-187) does not hold those bytes -- 'écar' is no word 'car'
    there, while bm()'s isalnum() test at k = 0 (sgrep.c:750-756) takes it.  agh_query_pattern follows suit."""
    lines = [b"car one", "écar two".encode("utf-8"), b"a car", b"xcar", "caré".encode("utf-8"), b"car\xe9 latin",
             b"\xe9car latin", b"scar", b"-car-", b"(car)", b"well-known fact", "unwell-known".encode(), b"a well-knowny",
             "éwell-known".encode("utf-8"), b"well-known\xc3\xa9", b"xwell y", b"a well b", "éwell".encode("utf-8"),
             b"car", b"ca", b"wellknown", b"we ll known"]
    text = b"\n".join(lines) + b"\n"
    p = tmp_path / "w.txt"
    p.write_bytes(text)
    args = ["-V0", "-w"] + (["-%d" % k] if k else []) + [pattern.decode(), str(p)]
    ref = subprocess.run([REF] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    want = [ln for ln in ref.stdout.split(b"\n") if ln]
    with agh.Query.pattern(pattern, k, word=True) as q:
        res, ms = q.scan_buffer(np.frombuffer(text, dtype=np.uint8), cap=1000)
    got = [text[s:e] for s, e, _ in ms]
    assert got == want, (pattern, k, got, want, ref.stderr[:200])
