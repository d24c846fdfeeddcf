"""agrep-hip on a box WITHOUT a GPU: what it decides on the host (option conflicts, patterns the library does not
compile: agh_compile_pattern is host-only) comes out with the reference's wording and exit status 2, and a scan
fails loudly -- there is no CPU scan engine to fall back to."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "agrep_amd", "agrep-hip")


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI) or os.path.getmtime(CLI) < os.path.getmtime(os.path.join(ROOT, "agrep_amd", "host", "agrep_hip.c")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "agrep_amd", "host")])
    return CLI


def _run(exe, args):
    p = subprocess.run([exe] + args, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr.decode("latin1")


@pytest.mark.parametrize("args,needle", [
    (["-c", "ab*c"], "regular expressions"),                      # regex: the reference's own engine, not this library
    (["-c", "a|b"], "regular expressions"),
    (["-c", "ab[cd"], "unmatched '[', ']'"),                       # maskgen.c's message
    (["-c", "a<bc"], "unmatched '<', '>'"),
    (["-c", "a;b,c"], "cannot handle OR (',') and AND (';')"),
    (["-c", "a" * 31 + "."], "pattern too long"),
    (["-w", "-x", "-c", "abc"], "illegal option combination (-x and -w)"),      # agrep.c:2188-2196
    (["-x", "-d", ";;", "-c", "abc"], "-d and -x are not compatible"),          # compat.c:89-96
    (["-B", "-c", "a[bc]d"], "-B needs a literal pattern"),
    (["-9", "-c", "abc"], "maximum number of errors"),
    (["-G", "abc"], "outside the GPU hot path"),
    (["--gpus", "0", "-c", "abc"], "--gpus needs a device count"),
])
def test_host_side_refusals(cli, args, needle):
    rc, out, err = _run(cli, args + [os.path.join(ROOT, "README.md")])
    assert rc == 2 and out == b"" and needle in err, (args, rc, err)


def test_a_scan_without_a_device_fails_loudly(cli):
    import agrep_amd
    if agrep_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    for args in (["-c", "agrep"], ["-1", "agrep"], ["-c", "agr[e]p"], ["-c", "ag#ep"]):
        rc, out, err = _run(cli, args + [os.path.join(ROOT, "README.md")])
        assert rc == 2 and out == b"" and "no usable HIP device" in err and "no CPU scan engine" in err, (args, err)
