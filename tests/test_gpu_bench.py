"""bench.py contract: one JSON line with the required fields, and the multi-rank control flow
(barriers, count reduction, MAX of the elapsed times) with two ranks sharing the one GPU of
the test box (gloo for the 16-byte count reduction; the real run uses nccl = RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _last_json(out):
    for line in reversed(out.decode().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError(out[-2000:])


def test_bench_single_and_two_ranks():
    env = dict(os.environ)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--total-gib", "0.5", "--steps", "3",
                          "--warmup", "1", "--cpu-sample-gib", "0.125"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    a = _last_json(one.stdout)
    for k in REQUIRED:
        assert k in a, k
    assert a["n_gpus"] == 1 and a["scaling"] == "strong" and a["dtype"] == "u8" and a["vs_baseline"] is None
    assert a["matched_equals_planted"] is True and a["k0"]["matched_records"] == a["k0"]["planted_records"]
    # traffic is measured in the run (rocprofv3 FETCH_SIZE pass of a child process) or null
    tr = a["roofline"]["traffic"]
    assert tr is None or 0.9 < tr / a["roofline"]["algorithmic_bytes_per_launch"] < 1.5, a["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(a["roofline"])
    assert a["roofline"]["bound"] == "hbm" and 0 < a["roofline"]["frac"] < 1
    cb = a["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] in ("reference", "port")
    if cb["kind"] == "reference":
        assert cb["count_equals_gpu"] is True
        # the reference over EVERY shard of the corpus (here 4 x 128 MiB), count by count
        assert cb["all_shards_count_equals_gpu"] is True, cb.get("all_shards")
        assert cb["all_shards"]["shards"] == 4 and cb["all_shards"]["gpu_count"] == a["matched_records"]
        # ... and the match SET: sha256 of the reference's printed lines == sha256 of the records the GPU returns
        assert cb["all_shards_records_sha256_equal"] is True, cb.get("all_shards")
        assert cb["all_shards"]["records_sha256_shards_equal"] == 4
    assert a["c2_records"]["matched_equals_planted"] is True
    if cb["kind"] == "reference":
        # the asearch() path at full size: -i on mixed-case shards, and the longest pattern the reference accepts
        assert cb["all_shards_nocase_records_sha256_equal"] is True, cb.get("all_shards_nocase", cb.get("all_shards_nocase_error"))
        assert cb["all_shards_nocase"]["shards"] == 4 and cb["all_shards_nocase"]["reference_count"] > 0
        m29 = a["c3"]["pinned_m29"]
        assert m29.get("records_sha256_equal") is True, m29
        assert m29["matched_records"] >= m29["planted_records_0_3_edits"] > 0
    # the engines behind the filter: every case measured, its slice count equal to the oracle's
    eng = a["engines"]
    cases = [k for k in eng if k != "workload"]
    assert len(cases) >= 15, cases
    for k in cases:
        assert eng[k].get("slice_equals_oracle") is True, (k, eng[k])
        assert 0 < eng[k]["roofline"]["frac"] < 1
    assert eng["m40_nocase_k6_fullscan_64bit_words"]["engine"].startswith("fullscan")

    env.update(AGH_BENCH_BACKEND="gloo", AGH_BENCH_ONE_GPU="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--total-gib", "0.5", "--steps", "3", "--warmup", "1", "--c5-file-mib", "64"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    b = _last_json(two.stdout)
    # configs[4] as a file job over the ranks: the -l hit vector reduced over both ranks (here: gloo)
    cf = b["c5_files"]
    assert cf.get("files") == 32 and cf["gpus_with_files"] == 2, cf
    for leg in ("every_byte", "dash_l"):
        assert cf[leg]["vector_equals_torch_all_reduce"] is True and len(cf[leg]["rank_ms_per_step"]) == 2
    assert cf["every_byte"]["files_listed"] == cf["dash_l"]["files_listed"] >= 8
    assert b["n_gpus"] == 2 and b["cpu_baseline"] is None
    # strong scaling: the same 0.5 GiB job, half of it per rank
    assert b["matched_records"] == a["matched_records"] and b["matched_equals_planted"] is True
    assert b["config"]["bytes_per_gpu"] * 2 == a["config"]["bytes_per_gpu"]
    assert b["config"]["total_bytes"] == a["config"]["total_bytes"] and b["scaling"] == "strong"
    # every rank's own step time and kernel rate, max / min over the ranks
    assert [r["rank"] for r in b["ranks"]] == [0, 1] and all(r["bytes"] == b["config"]["bytes_per_gpu"] for r in b["ranks"])
    assert b["rank_ms_per_step_min"] <= b["rank_ms_per_step_max"] and abs(b["rank_ms_per_step_max"] - b["ms_per_step"]) < 1e-3
    assert all(0 < r["roofline"]["frac"] < 1 for r in b["ranks"]) and len(b["k0"]["ranks"]) == 2


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment starts the two ranks itself
    (torch.distributed.run, one process per GPU) -- here both on GPU 0 with the counts over gloo."""
    env = dict(os.environ, AGH_BENCH_BACKEND="gloo", AGH_BENCH_ONE_GPU="1")
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total-gib", "0.5",
                        "--steps", "3", "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    b = _last_json(r.stdout)
    assert b["n_gpus"] == 2 and b["matched_equals_planted"] is True and b["scaling"] == "strong"
    assert b["config"]["bytes_per_gpu"] * 2 == b["config"]["total_bytes"]


def test_bench_rccl_code_path_with_one_rank():
    """The nccl branch of bench.py end to end on the one GPU there is: torch.distributed over RCCL,
    rank 0's ncclUniqueId broadcast, the C-ABI's own communicator (agh_comm_init_rank) and
    agh_scan_device_reduce (scan + ncclAllReduce on the scan's stream, one host sync) in every timed step --
    with a world of one rank."""
    env = dict(os.environ, AGH_BENCH_FORCE_DIST="1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--total-gib", "0.5", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-traffic", "--no-engines", "--no-configs", "--c5-file-mib", "64"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    a = _last_json(r.stdout)
    # the file job's hit vector through agh_reduce_file_hits on RCCL (a communicator of one rank)
    cf = a["c5_files"]
    assert cf.get("hit_reduction", "").startswith("agh_reduce_file_hits") and cf["rccl_ranks"] == 1, cf
    assert cf["every_byte"]["vector_equals_torch_all_reduce"] is True and cf["dash_l"]["files_listed"] == cf["every_byte"]["files_listed"]
    assert a["config"]["count_reduction"].startswith("agh_scan_device_reduce")
    assert a["matched_equals_planted"] is True and a["n_gpus"] == 1
    assert a["rccl_ranks"] == 1 and a["ranks"][0]["rccl_ranks"] == 1      # what agh_comm_info says about the communicator


def test_scan_device_reduce_one_rank():
    """agh_scan_device_reduce on a communicator of one rank: the totals are the scan's own counts -- on the
    fused count-only kernel (device-side sum + ncclAllReduce on the scan's stream), on a text whose count-only
    scan gives up (a 3 MiB record in front of a match: rerun, second all-reduce), on the one-pass -f kernel
    and on a query without a count-only pipeline of its own (host-side all-reduce of the same shape)."""
    import torch
    import numpy as np
    import agrep_amd as A
    import _oracle as O
    comm = A.Comm(A.Comm.unique_id(), 1, 0)
    try:
        text, planted = O.corpus(2048, seed=3, variants=O.VARIANTS_C2, plant_period=40)
        want = O.asearch(O.PATTERN_C2, 2, text)[0]
        t = torch.from_numpy(text).cuda()
        with A.Query(O.PATTERN_C2, 2) as q:
            res, tot = q.scan_device_reduce(comm, t.data_ptr(), t.numel())
            assert res.n_matched == tot[0] == want and res.fused_segments == 1
            # a record longer than the look-back of the count-only verifier, a match at its end
            long = np.concatenate([np.full(3 << 20, ord("x"), dtype=np.uint8), np.frombuffer(b" approximatematch\n", dtype=np.uint8), text])
            tl = torch.from_numpy(long).cuda()
            res2, tot2 = q.scan_device_reduce(comm, tl.data_ptr(), tl.numel())
            assert res2.lean_reruns == 0 and res2.n_matched == tot2[0] == want + 1     # (resolved on the device)
            os.environ["AGH_GIVEUP_CAP"] = "0"                      # ... without the give-up list: the rerun
            try:
                res2b, tot2b = q.scan_device_reduce(comm, tl.data_ptr(), tl.numel())
            finally:
                del os.environ["AGH_GIVEUP_CAP"]
            assert res2b.lean_reruns == 1 and res2b.n_matched == tot2b[0] == want + 1
            res3, tot3 = q.scan_device_reduce(comm, t.data_ptr(), t.numel(), flags=A.COUNT | A.FORCE_NUMBERED)
            assert res3.n_matched == tot3[0] == want and tot3[1] == res3.n_records > 0
        pats = [b"approxim", b"matematch", b"zzzzqqqq"]
        with A.Query.multi(pats, k=1) as qm:
            r4, tot4 = qm.scan_device_reduce(comm, t.data_ptr(), t.numel())
            r5 = qm.scan_device(t.data_ptr(), t.numel(), flags=A.COUNT | A.FORCE_NUMBERED)
            assert r4.fused_segments == 1 and r4.n_matched == tot4[0] == r5.n_matched > 0
    finally:
        comm.close()
