"""The C host CLI (agrep_amd/agrep-hip) against the reference CLI (oracle/_ref/agrep) on the
same files: stdout, stderr shape and exit status for the hot-path option surface
(-# -c -l -i -n -h -s -d -B -y -V0), SURVEY.md Appendix A."""
import os
import subprocess

import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "agrep_amd", "agrep-hip")
REF = os.path.join(O.REF_DIR, "agrep")


def _run(exe, args, stdin=None, env=None):
    p = subprocess.run([exe] + args, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, **env) if env else None)
    return p.returncode, p.stdout, p.stderr


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "agrep_amd", "host")])
    d = tmp_path_factory.mktemp("cli")
    out = []
    for i, (pages, period) in enumerate(((24, 30), (8, 11), (4, 1000000))):
        text, _ = O.corpus(pages, seed=100 + i, variants=O.VARIANTS_C2, plant_period=period)
        p = d / ("f%d.txt" % i)
        p.write_bytes(text.tobytes())
        out.append(str(p))
    return out


needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/agrep not built")


@needs_ref
@pytest.mark.parametrize("args", [
    ["-V0", "-2", "-c"], ["-2", "-c"], ["-V0", "-2"], ["-2"], ["-V0", "-i", "-2"],
    ["-V0", "-1", "-l"], ["-2", "-l"], ["-V0", "-2", "-h"], ["-V0", "-i", "-n", "-2"],
    ["-V0", "-c"], ["-V0", "-s", "-2"], ["-V0", "-3", "-ci"], ["-V0", "-ic", "-2"],
    ["-V0", "-I2", "-c", "-2"], ["-V0", "-D2", "-S2", "-2"], ["-V0", "-I3", "-D3", "-c", "-3"],
])
def test_cli_matches_reference(files, args):
    for fl in (files[:1], files):
        a = args + ["approximatematch"] + fl
        rc_r, out_r, err_r = _run(REF, a)
        rc_g, out_g, err_g = _run(CLI, a)
        assert out_g == out_r, (a, out_g[:300], out_r[:300])
        assert rc_g == rc_r, a
        assert err_g == b""


@needs_ref
@pytest.mark.parametrize("args", [["-V0", "-i", "-v", "-2", "-c"], ["-V0", "-i", "-v", "-2"],
                                  ["-V0", "-i", "-v", "-n", "-1"], ["-V0", "-i", "-v", "-l", "-2"],
                                  ["-V0", "-i", "-vc", "-1"]])
def test_cli_inverse_matches_reference(files, args):
    """-v on the asearch.c path (-i with errors keeps the reference off sgrep.c, whose engines
    ignore -v: `agrep -v -c pat` == `agrep -c pat` there)."""
    for fl in (files[1:2], files):
        a = args + ["approximatematch"] + fl
        rc_r, out_r, err_r = _run(REF, a)
        rc_g, out_g, err_g = _run(CLI, a)
        assert out_g == out_r, (a, out_g[:300], out_r[:300])
        assert rc_g == rc_r, a


@needs_ref
def test_cli_stdin_and_missing_file(files):
    data = open(files[1], "rb").read()
    rc_r, out_r, _ = _run(REF, ["-V0", "-2", "-c", "approximatematch", "/dev/stdin"], stdin=data)
    rc_g, out_g, _ = _run(CLI, ["-V0", "-2", "-c", "approximatematch", "/dev/stdin"], stdin=data)
    assert (rc_g, out_g) == (rc_r, out_r)
    rc_g, out_g, _ = _run(CLI, ["-V0", "-2", "-c", "approximatematch"], stdin=data)   # plain stdin
    assert (rc_g, out_g) == (rc_r, out_r)
    a = ["-V0", "-2", "-c", "approximatematch", files[0], "/nonexistent/x", files[1]]
    rc_r, out_r, err_r = _run(REF, a)
    rc_g, out_g, err_g = _run(CLI, a)
    assert out_g == out_r and rc_g == rc_r
    assert b"no such file or directory" in err_g and b"no such file or directory" in err_r


@needs_ref
def test_cli_best_match(files, tmp_path):
    p = tmp_path / "words.txt"
    p.write_bytes(b"alpha\nhomogeneous\nbeta\nhomogenous\ngamma\n")
    for pat in ("homogenoss", "homogeneous", "zzzzqqqq"):
        a = ["-V0", "-B", "-y", pat, str(p)]
        rc_r, out_r, err_r = _run(REF, a)
        rc_g, out_g, err_g = _run(CLI, a)
        assert out_g == out_r, (pat, out_g, out_r)
        # stderr: "<prog>: N word(s) match(es) within D error(s)" -- program names differ
        assert err_g.split(b":", 1)[-1] == err_r.split(b":", 1)[-1], (err_g, err_r)
        assert rc_g == rc_r


@needs_ref
def test_cli_single_byte_delimiter_counts(tmp_path):
    p = tmp_path / "semi.txt"
    p.write_bytes(b"one approximatematch;two;three aproximatematch;four apprximatemtch;five")
    for k in (0, 1, 2):
        a = ["-V0", "-i", "-%d" % k, "-c", "-d", ";", "approximatematch", str(p)] if k else \
            ["-V0", "-i", "-c", "-d", ";", "approximatematch", str(p)]
        rc_r, out_r, _ = _run(REF, a)
        rc_g, out_g, _ = _run(CLI, a)
        assert (rc_g, out_g) == (rc_r, out_r), (k, out_g, out_r)
    # records under -d: the delimiter in front of every record but the first, nothing behind
    # (output(), agrep.c:3805-3956, through the asearch path); -n numbers in front of the delimiter
    for a in (["-V0", "-i", "-2", "-d", ";"], ["-V0", "-i", "-n", "-2", "-d", ";"], ["-V0", "-i", "-1", "-d", ";"]):
        a = a + ["approximatematch", str(p)]
        rc_r, out_r, _ = _run(REF, a)
        rc_g, out_g, _ = _run(CLI, a)
        assert (rc_g, out_g) == (rc_r, out_r), (a, out_g, out_r)
        assert rc_g == 0 or b";three aproximatematch" in out_g


@needs_ref
def test_cli_multi_byte_delimiter_counts(tmp_path):
    """-d 'From ' (mbox) and -d '$$' (paragraphs): counts equal the reference's asearch path."""
    mbox = (b"From alice\nsubject: approximate matching\nbody\n"
            b"From bob\nnothing here\n\nFrom carol\napproximatematch is here\n"
            b"From dave\naproximatemach twice removed\n")
    para = b"para one\nline\n\npara two approximatematch\n\n\npara three aproximatematc\nx\n\n"
    for text, dl in ((mbox, "From "), (para, "$$")):
        p = tmp_path / "d.txt"
        p.write_bytes(text)
        for k in (0, 1, 2):
            a = ["-V0", "-i"] + (["-%d" % k] if k else []) + ["-c", "-d", dl, "approximatematch", str(p)]
            rc_r, out_r, _ = _run(REF, a)
            rc_g, out_g, _ = _run(CLI, a)
            assert (rc_g, out_g) == (rc_r, out_r), (dl, k, out_g, out_r)
        a = ["-V0", "-i", "-1", "-d", dl, "approximatematch", str(p)]       # the records themselves
        rc_r, out_r, _ = _run(REF, a)
        rc_g, out_g, _ = _run(CLI, a)
        assert (rc_g, out_g) == (rc_r, out_r), (dl, out_g, out_r)


@needs_ref
@pytest.mark.parametrize("args", [
    ["-V0", "-c", "-1", "appr[ox]ximatematch"], ["-V0", "-2", "approx[a-m]matematch"], ["-V0", "-i", "-2", "[^b-z]PProximatematch"],
    ["-V0", "-c", "approx#match"], ["-V0", "-1", "-n", "appr#mate#ch"], ["-V0", "-c", "-1", "approxi;matematch"],
    ["-V0", "-1", "aproxi,matemmat"], ["-V0", "-c", "^appro"], ["-V0", "-1", "-c", "tematch$"],
    ["-V0", "-2", "-c", "<appro>ximatematch"], ["-V0", "-1", "-c", "appro.imatematch"], ["-V0", "-c", "-w", "m[a-c]tch"],
    ["-V0", "-1", "-l", "appr[ox]ximatematch"], ["-2", "-c", "a\\.proximatematch"],
])
def test_cli_pattern_language_matches_reference(files, args):
    """classes, '.', '#', <exact>, ^ $, ';' and ',' lists: compiled by the library (agh_query_pattern) and by the
    reference's preprocess() + maskgen() -- same bytes on stdout, same exit status"""
    for fl in (files[:1], files):
        rc_r, out_r, _ = _run(REF, args + fl)
        rc_g, out_g, err_g = _run(CLI, args + fl)
        assert (rc_g, out_g) == (rc_r, out_r), (args, out_g[:300], out_r[:300], err_g[:200])


def test_cli_rejects_what_is_outside_the_hot_path(files):
    for a in (["-2", "a|b", files[0]], ["-2", "ab*cdefgh", files[0]], ["-2", "[ab.]cdefgh", files[0]], ["-G", "x", files[0]],
              ["-B", "[ab]cdefgh", files[0]], ["-x", "-d", ";;", "abc", files[0]],
              ["-9", "approximatematch", files[0]], ["-2", "ab", files[0]]):
        rc, out, err = _run(CLI, a)
        assert rc == 2 and err


def _n_devices():
    import agrep_amd
    return agrep_amd.device_count()


@pytest.mark.parametrize("args", [["-V0", "-2", "-c"], ["-2", "-c"], ["-V0", "-2"], ["-V0", "-n", "-i", "-2"],
                                  ["-V0", "-1", "-l"], ["-2", "-l"], ["-V0", "-c"], ["-V0", "-h", "-2"],
                                  ["-V0", "-i", "-v", "-l", "-2"], ["-V0", "-v", "-l"], ["-V0", "-i", "-v", "-c", "-1"]])
def test_cli_multi_gpu_equals_single(files, args):
    """--gpus N: every file cut into N record-aligned shards, one host thread + one
    query per device; the -c sum / the -l hit vector are host additions over the threads of the one
    process (no communicator), and with AGH_CLI_RCCL=1 also reduced with RCCL inside the C-ABI
    (agh_reduce_counts_all / agh_reduce_file_hits_all: ncclCommInitAll + ncclAllReduce, also with
    N = 1) and compared.  Output, order and exit status equal the one-GPU run."""
    n_dev = _n_devices()
    for g in sorted({1, min(2, n_dev), n_dev}):
        for fl in (files[:1], files):
            a = args + ["approximatematch"] + fl
            rc_1, out_1, err_1 = _run(CLI, a)
            for env in ((None, {"AGH_CLI_RCCL": "1"}) if fl is not files else (None,)):
                rc_g, out_g, err_g = _run(CLI, ["--gpus", str(g)] + a, env=env)
                assert err_g == b"", err_g[:500]
                assert out_g == out_1, (g, a, out_g[:300], out_1[:300])
                assert rc_g == rc_1


@pytest.mark.parametrize("args", [["-V0", "-2"], ["-V0", "-n", "-i", "-2"], ["-V0", "-2", "-c"], ["-V0", "-v", "-n", "-2"],
                                  ["-V0", "-d", "e ", "-n", "-1"], ["-V0", "-d", "q", "-n", "-i", "-1"], ["-V0", "-h", "-1"]])
def test_cli_shards_print_in_file_order_while_scanning(files, args, tmp_path):
    """--gpus N record output: every shard streams through agh_scan_fd_range_emit on a thread of its own; the shard
    whose turn it is prints straight from its emit() calls, the ones behind it hold their records back until the
    shards in front are done (file order, -n numbers offset by the records in front).  AGH_CLI_SHARE_DEVICES=1 puts
    the N threads on the box's one GPU, so N = 2, 3, 5 run here; AGH_STREAM_SEG_MB=1 gives every shard several
    emit() calls."""
    for fl in (files[:1], files, files + ["/nonexistent/x"] + files[:1]):
        a = args + ["approximatematch"] + fl
        rc_1, out_1, err_1 = _run(CLI, a)
        for g in (2, 3, 5):
            rc_g, out_g, err_g = _run(CLI, ["--gpus", str(g)] + a, env={"AGH_CLI_SHARE_DEVICES": "1", "AGH_STREAM_SEG_MB": "1"})
            assert out_g == out_1, (g, a, out_g[:300], out_1[:300], err_g[:300])
            assert rc_g == rc_1 and err_g == err_1


def test_cli_multi_gpu_pattern_file(files, tmp_path):
    pf = tmp_path / "pats.txt"
    pf.write_bytes(b"approximatematch\naproximatematch\nzzzzqqqq\n")
    for mode in (["-c"], ["-l"], [], ["-n"]):
        a = ["-V0"] + mode + ["-f", str(pf)] + files
        rc_1, out_1, _ = _run(CLI, a)
        rc_g, out_g, err_g = _run(CLI, ["--gpus", "1"] + a)
        assert err_g == b"" and out_g == out_1 and rc_g == rc_1, (mode, out_g[:200], out_1[:200])
        # ... and over three shards per file on the one GPU (records: printed shard after shard while the others scan)
        rc_3, out_3, err_3 = _run(CLI, ["--gpus", "3"] + a, env={"AGH_CLI_SHARE_DEVICES": "1", "AGH_STREAM_SEG_MB": "1"})
        assert err_3 == b"" and out_3 == out_1 and rc_3 == rc_1, (mode, out_3[:200], out_1[:200])
    # -f with one error (--approx-f) on a dense set: the tile kernel's numbered form behind every shard's record list
    pd = tmp_path / "dense.txt"
    pd.write_bytes(b"appr\nmatch\natema\nzqzq\n")
    for mode in (["-c"], [], ["-n"]):
        a = ["-V0", "--approx-f", "-1"] + mode + ["-f", str(pd)] + files[:2]
        rc_1, out_1, err_1 = _run(CLI, a)
        assert err_1 == b"" and out_1
        rc_3, out_3, err_3 = _run(CLI, ["--gpus", "3"] + a, env={"AGH_CLI_SHARE_DEVICES": "1", "AGH_STREAM_SEG_MB": "1"})
        assert err_3 == b"" and out_3 == out_1 and rc_3 == rc_1, (mode, out_3[:200], out_1[:200])


def test_cli_q6_divergence_is_deliberate(tmp_path):
    """Quirk Q6 (sgrep.c:226-236: TR[] folds ASCII case unconditionally on the k = 0 bm() path):
    the reference's `agrep hello` also prints "Hello World".  This build does not reproduce it:
    without -i a pattern is case-sensitive at every k (as the reference itself is at k > 0 and on
    its bitap path), with -i both agree."""
    f = tmp_path / "mixed.txt"
    f.write_bytes(b"hello world\nHello World\nHELLO\nnothing here\n")
    rc, out, err = _run(CLI, ["-V0", "hello", str(f)])
    assert out == b"hello world\n" and err == b""
    rc, out, err = _run(CLI, ["-V0", "-i", "hello", str(f)])
    assert out == b"hello world\nHello World\nHELLO\n"
    if os.path.exists(REF):
        _, out_r, _ = _run(REF, ["-V0", "hello", str(f)])
        assert out_r == b"hello world\nHello World\nHELLO\n"          # Q6, stated not emulated
        _, out_n, _ = _run(REF, ["-V0", "-n", "hello", str(f)])          # bitap path: case-sensitive
        assert out_n == b"1: hello world\n"
        _, out_ri, _ = _run(REF, ["-V0", "-i", "hello", str(f)])
        assert out_ri == out


@needs_ref
def test_cli_nocase_letter_delimiter(tmp_path):
    """-i -d q ('Q' ends a record too, maskgen.c:259-266): counts equal the reference's (record
    placement under -d is the reference's own output(): tests/test_gpu_refshim.py)."""
    t, _ = O.corpus(16, seed=9, variants=O.VARIANTS_C2, plant_period=7, upper_permille=300)
    f = tmp_path / "dq.txt"
    f.write_bytes(t.tobytes().replace(b"\n", b"q"))
    for args in (["-V0", "-i", "-d", "q", "-2", "-c"], ["-V0", "-i", "-d", "q", "-1", "-c"], ["-V0", "-i", "-d", "q", "-c"]):
        a = args + ["approximatematch", str(f)]
        rc_r, out_r, _ = _run(REF, a)
        rc_g, out_g, err_g = _run(CLI, a)
        assert (rc_g, out_g) == (rc_r, out_r), (a, out_g[:200], out_r[:200], err_g[:200])


@needs_ref
def test_cli_config_c1_exact_m8_1mib_count(tmp_path):
    """BASELINE configs[0] as worded: exact m = 8 pattern, 1 MiB ASCII file, k = 0, -c -- the bm() path
    of the reference (sgrep.c:694-1016) against the device's k = 0 filter path, CLI to CLI."""
    text, planted = O.corpus(256, seed=8, variants=(b"approxim", b"approxim", b"aproxim"), plant_period=20)
    f = tmp_path / "c1.txt"
    f.write_bytes(text.tobytes())
    assert f.stat().st_size == 1 << 20
    for args in (["-c"], ["-V0", "-c"], ["-V0", "-l"], ["-V0"]):
        a = args + ["approxim", str(f)]
        rc_r, out_r, _ = _run(REF, a)
        rc_g, out_g, err_g = _run(CLI, a)
        assert (rc_g, out_g) == (rc_r, out_r) and err_g == b"", (a, out_g[:200], out_r[:200])
    assert int(_run(CLI, ["-V0", "-c", "approxim", str(f)])[1]) >= planted[0] + planted[1] > 0


@needs_ref
def test_cli_word_and_line_guards(files, tmp_path):
    """-w / -x in the C CLI (agh_query_literal_ex): k = 0 against the reference's simple path, k > 0
    against its maskgen path."""
    for w in ("the", "at", "match"):
        for mode in (["-V0", "-w", "-c"], ["-V0", "-w"], ["-V0", "-w", "-l"], ["-V0", "-w", "-1", "-c"],
                     ["-V0", "-w", "-i", "-1"]):
            a = mode + [w] + files[:2]
            rc_r, out_r, _ = _run(REF, a)
            rc_g, out_g, err_g = _run(CLI, a)
            assert (rc_g, out_g) == (rc_r, out_r), (a, out_g[:200], out_r[:200], err_g[:200])
    lines = open(files[1], "rb").read().split(b"\n")[:300]
    for i in range(7, len(lines), 41):
        lines[i] = b"approximatematch"
    f = tmp_path / "x.txt"
    f.write_bytes(b"\n".join(lines) + b"\n")
    for mode in (["-V0", "-x", "-c"], ["-V0", "-x", "-l"], ["-V0", "-x", "-n"], ["-V0", "-x", "-1", "-c"]):
        a = mode + ["approximatematch", str(f)]
        assert _run(CLI, a)[:2] == _run(REF, a)[:2], a


def test_cli_pattern_file_with_errors(files, tmp_path):
    """--approx-f: -# applies to the patterns of -f (BASELINE config 5; the reference warns and ignores it,
    compat.c:34-37).  Same count as the union of the single-pattern scans through the same CLI."""
    pf = tmp_path / "pats.txt"
    pats = [b"approximatematch", b"zzzzqqqqxx", b"etaoinshrdlu"]
    pf.write_bytes(b"\n".join(pats) + b"\n")
    rc, out, err = _run(CLI, ["--approx-f", "-V0", "-2", "-c", "-f", str(pf), files[0]])
    assert err == b"", err
    single = int(_run(CLI, ["-V0", "-2", "-c", "approximatematch", files[0]])[1])
    assert int(out) >= single > 0
    rc2, out2, err2 = _run(CLI, ["-V0", "-2", "-c", "-f", str(pf), files[0]])      # reference behaviour: warning, exact
    assert b"not supported with -f" in err2 and int(out2) <= int(out)
    for g in ("1",):
        assert _run(CLI, ["--gpus", g, "--approx-f", "-V0", "-2", "-l", "-f", str(pf)] + files)[1] == \
            _run(CLI, ["--approx-f", "-V0", "-2", "-l", "-f", str(pf)] + files)[1]


def test_cli_refuses_what_the_library_cannot_honour(files, tmp_path):
    """Option combinations that must not be dropped silently: -w with -x (the reference: "illegal option
    combination", agrep.c:2188-2196), -w / -x with --approx-f and errors (the guards need a verbatim
    occurrence), --gpus with -i and a multi-byte letter delimiter (shards are cut on raw bytes)."""
    pf = tmp_path / "p.txt"
    pf.write_bytes(b"approximatematch\n")
    for a in (["-w", "-x", "-c", "match", files[0]],
              ["--approx-f", "-1", "-w", "-c", "-f", str(pf), files[0]],
              ["--approx-f", "-2", "-x", "-l", "-f", str(pf), files[0]],
              ["--gpus", "1", "-i", "-d", "From ", "-c", "match", files[0]]):
        rc, out, err = _run(CLI, a)
        assert rc == 2 and out == b"" and err, a
    # ... while the same guards with an exact pattern file are served
    rc, out, err = _run(CLI, ["-V0", "-w", "-c", "-f", str(pf), files[0]])
    assert rc != 2 and err == b""
    import agrep_amd as A
    with pytest.raises(A.AghError):
        A.Query(b"match", 0, word=True, wholeline=True)
