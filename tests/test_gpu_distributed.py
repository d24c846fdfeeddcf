"""The N > 1 step of a sharded count-only scan (agh_scan_device_reduce) with TWO ranks on one GPU: the exchange runs
over gloo through a communicator on the caller's transport (agh_comm_init_custom) -- RCCL refuses two ranks on one
device, and the boxes these tests run on have one.  What is checked is what RCCL cannot change: every rank issues
the same sequence of collectives whatever path its own scan took (one all-reduce of (matched, records, gave-up);
a second one with the final counts only when SOME rank had to rerun a segment), and the totals are the sum of the
shards' oracle counts."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _texts(rank, case):
    text, _ = O.corpus(40 + 8 * rank, first_page=100 * rank, seed=12345, variants=O.VARIANTS_C2, plant_period=13)
    tb = text.tobytes()
    if case == "rerun" and rank == 1:
        # a record of 2 MiB in front of a match: with AGH_GIVEUP_CAP=0 the count-only scan gives the segment up
        # (its verifier looks back 1 MiB for the record start) and the host runs it again with record numbers
        tb = b"x" * (2 << 20) + b" approximatematch tail\n" + tb
    return tb


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import agrep_amd as A
        A.set_device(0)
        log = []

        def allreduce(vals, elem):
            t = torch.tensor(vals, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM if elem == 8 else dist.ReduceOp.MAX)
            log.append((len(vals), elem))
            return t.tolist()
        comm = A.Comm.custom(allreduce, world, rank)
        info = comm.info()
        out = []
        for case in ("plain", "rerun"):
            tb = _texts(rank, case)
            if case == "rerun":
                os.environ["AGH_GIVEUP_CAP"] = "0"
            dev = torch.frombuffer(bytearray(tb + b"\0" * 64), dtype=torch.uint8).cuda()
            with A.Query(O.PATTERN_C2, 2) as qq:
                res, tot = qq.scan_device_reduce(comm, dev.data_ptr(), len(tb), flags=A.COUNT)
            out.append((case, int(tot[0]), int(res.n_matched), int(res.lean_reruns), list(log)))
            del log[:]
        hits = comm.reduce_file_hits([rank == 0, False, rank == 1])
        q.put((rank, info, out, hits))
        comm.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_issue_the_same_collectives_when_one_reruns_a_segment():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = {case: [O.asearch(O.PATTERN_C2, 2, _texts(r, case))[0] for r in range(world)] for case in ("plain", "rerun")}
    for rank, info, out, hits in outs:
        assert info["nranks"] == world and info["rank"] == rank
        assert hits == [True, False, True]
        for case, total, mine, reruns, log in out:
            assert mine == want[case][rank] and total == sum(want[case]), (rank, case)
            if case == "plain":
                assert reruns == 0 and log == [(3, 8)], (rank, log)
            else:
                # rank 1 reran its segment; BOTH ranks reduce a second time (the final counts)
                assert reruns == (1 if rank == 1 else 0) and log == [(3, 8), (2, 8)], (rank, reruns, log)
