"""N > 1 path on CPU: two gloo ranks shard a corpus, scan their shards (the oracle stands in
for the device scan here -- no GPU in this test) and reduce the counts; the result must equal
the single-process answer."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _oracle as O  # noqa: E402
from agrep_amd import shard  # noqa: E402


def _worker(rank, world, port, total_pages, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (1) page-sharded synthetic corpus, as bench.py does it
        first, count = shard.shard_pages(total_pages, world, rank)
        text, _ = O.corpus(count, first_page=first, seed=12345, variants=O.VARIANTS_C2,
                           plant_period=17)
        n, recs = O.asearch(O.PATTERN_C2, 2, text)[0], int((text == 10).sum())
        tot, trec = shard.reduce_counts(n, recs)
        # (2) a ragged host buffer cut at record boundaries
        rng = np.random.default_rng(5)
        buf, _ = O.corpus(9, seed=77, variants=O.VARIANTS_C2, plant_period=3)
        buf = buf[: buf.size - int(rng.integers(1, 300))].copy()      # no trailing newline
        cuts = shard.record_cuts(buf, world)
        piece = buf[cuts[rank]:cuts[rank + 1]]
        tot2, _ = shard.reduce_counts(O.asearch(O.PATTERN_C2, 2, piece)[0])
        hits = shard.reduce_file_hits([rank == 0, False, rank == 1])
        q.put((rank, tot, trec, tot2, hits, cuts))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_reduce():
    world, total_pages = 2, 23
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_pages, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole, _ = O.corpus(total_pages, seed=12345, variants=O.VARIANTS_C2, plant_period=17)
    want = O.asearch(O.PATTERN_C2, 2, whole)[0]
    buf, _ = O.corpus(9, seed=77, variants=O.VARIANTS_C2, plant_period=3)
    rng = np.random.default_rng(5)
    buf = buf[: buf.size - int(rng.integers(1, 300))].copy()
    want2 = O.asearch(O.PATTERN_C2, 2, buf)[0]
    assert want > 0 and want2 > 0
    for rank, tot, trec, tot2, hits, cuts in outs:
        assert tot == want
        assert trec == int((whole == 10).sum())
        assert tot2 == want2
        assert hits == [True, False, True]
        assert cuts[0] == 0 and cuts[-1] == buf.size


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_record_cuts_partition_records(world):
    rng = np.random.default_rng(world)
    for trial in range(20):
        n = int(rng.integers(0, 5000))
        a = rng.integers(97, 100, size=n).astype(np.uint8)
        a[rng.random(n) < 0.03] = 10
        cuts = shard.record_cuts(a, world)
        assert cuts[0] == 0 and cuts[-1] == n and all(x <= y for x, y in zip(cuts, cuts[1:]))
        for c in cuts[1:-1]:
            assert c == n or c == 0 or a[c - 1] == 10
        pieces = [a[cuts[r]:cuts[r + 1]].tobytes() for r in range(world)]
        assert b"".join(pieces) == a.tobytes()
        # every record is whole inside one piece: per-piece counts add up
        pat = b"abca"
        total = sum(O.dp_count(pat, 1, p)[0] for p in pieces)
        assert total == O.dp_count(pat, 1, a.tobytes())[0]


def test_shard_pages_cover_everything():
    for total in (0, 1, 7, 64, 1048576):
        for world in (1, 2, 3, 8):
            got = [shard.shard_pages(total, world, r) for r in range(world)]
            assert got[0][0] == 0
            assert sum(c for _, c in got) == total
            for (f0, c0), (f1, _) in zip(got, got[1:]):
                assert f0 + c0 == f1


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_c_abi_shard_cuts_equal_python_rule(world, tmp_path):
    """agh_shard_cuts_fd (host-only code of the C-ABI, used by `agrep-hip --gpus N`) places the cuts
    exactly where shard.record_cuts does: files from empty to 200 kB, with and without a trailing
    newline, long records, delimiters right at the nominal offsets."""
    import agrep_amd
    rng = np.random.default_rng(100 + world)
    cases = [b"", b"\n", b"abc", b"abc\n", b"\n" * 50, b"x" * 5000, b"x" * 4999 + b"\n" + b"y" * 5000]
    for trial in range(25):
        n = int(rng.integers(0, 200000))
        a = rng.integers(97, 100, size=n).astype(np.uint8)
        a[rng.random(n) < (0.0005 if trial % 3 == 0 else 0.02)] = 10
        cases.append(a.tobytes())
    for i, data in enumerate(cases):
        f = tmp_path / ("c%d.bin" % i)
        f.write_bytes(data)
        fd = os.open(str(f), os.O_RDONLY)
        try:
            got = agrep_amd.shard_cuts_fd(fd, world)
        finally:
            os.close(fd)
        want = shard.record_cuts(np.frombuffer(data, dtype=np.uint8), world)
        assert got == want, (world, i, len(data), got, want)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_c_abi_shard_cuts_multi_byte_delimiter(world, tmp_path):
    """The same rule for delimiters of several bytes that cannot overlap themselves ("\\r\\n", "From "):
    inner cut r = end of the first delimiter occurrence that ends at or after size * r / world.
    Self-overlapping delimiters ("\\n\\n") are refused: which occurrences count depends on the scan's
    start (asearch.c:54-57)."""
    import agrep_amd
    rng = np.random.default_rng(7 + world)
    for delim in (b"\r\n", b"From ", b"ab"):
        for trial in range(8):
            n = int(rng.integers(0, 300000))
            a = rng.integers(97, 101, size=n).astype(np.uint8).tobytes()
            pieces = []
            at = 0
            while at < n:
                step = int(rng.integers(1, 4000 if trial % 2 else 200))
                pieces.append(a[at:at + step].replace(delim, b"zz"))
                at += step
            data = delim.join(pieces)
            f = tmp_path / ("m%d.bin" % trial)
            f.write_bytes(data)
            want = [0]
            for r in range(1, world):
                nominal = max(want[-1], len(data) * r // world)
                if nominal >= len(data):
                    want.append(len(data))
                    continue
                if nominal == 0:
                    want.append(0)
                    continue
                i = data.find(delim, max(0, nominal - len(delim)))
                want.append(len(data) if i < 0 else i + len(delim))
            want.append(len(data))
            fd = os.open(str(f), os.O_RDONLY)
            try:
                got = agrep_amd.shard_cuts_fd(fd, world, delim=delim)
            finally:
                os.close(fd)
            assert got == want, (delim, trial, got, want)
    f = tmp_path / "nn.bin"
    f.write_bytes(b"a\n\n\nb\n\n" * 100)
    fd = os.open(str(f), os.O_RDONLY)
    try:
        with pytest.raises(agrep_amd.AghError):
            agrep_amd.shard_cuts_fd(fd, 2, delim=b"\n\n")
    finally:
        os.close(fd)


def _files_worker(rank, world, port, n_files, q):
    """configs[4] as bench.py runs it at N > 1 (c5_file_hits_job): the files are dealt in blocks to the first G ranks,
    every rank of the group takes part in the MAX reduction of the hit vector, the ones without files with zeros"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G = min(world, 2)                       # (bench.py: min(N, 4); here: one rank of three has no files)
        per = n_files // G
        mine = list(range(rank * per, (rank + 1) * per)) if rank < G else []
        pats = [b"needle", b"haystack", b"zebra"]
        hits = [False] * n_files
        for f in mine:
            text, _ = O.corpus(2, first_page=2 * f, seed=55, variants=(b"a needle b", b"one haystack"),
                               plant_period=40 if f % 3 == 0 else 1 << 30)
            hits[f] = O.multi_exact_count(pats, text)[0] > 0
        q.put((rank, shard.reduce_file_hits(hits), mine))
    finally:
        dist.destroy_process_group()


def test_three_ranks_file_hit_vector_with_an_idle_rank():
    world, n_files = 3, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_files_worker, args=(r, world, port, n_files, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pats = [b"needle", b"haystack", b"zebra"]
    want = []
    for f in range(n_files):
        text, _ = O.corpus(2, first_page=2 * f, seed=55, variants=(b"a needle b", b"one haystack"),
                           plant_period=40 if f % 3 == 0 else 1 << 30)
        want.append(O.multi_exact_count(pats, text)[0] > 0)
    assert any(want) and not all(want)
    assert [m for _, _, m in outs] == [[0, 1, 2, 3], [4, 5, 6, 7], []]
    for _, vec, _ in outs:
        assert vec == want
