#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X agrep scanner.

Metric (BASELINE.json): GB/s scanned (+ Mmatches/s), m=16 pattern, k=2, synthetic
newline-delimited corpus resident in HBM.  A "step" is one complete -c scan of the rank's
shard (sweep + verify + count, through the C-ABI agh_scan_device) followed, for N > 1, by the
RCCL all-reduce of the per-rank counts.  Weak scaling: every rank owns `--gib` GiB (default 4 =
BASELINE configs[1]) of the same deterministic corpus (disjoint page ranges).

    python bench.py                     # 1 GPU, configs[1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel k_sweep<H>: algorithmic bytes (1 B per corpus byte) / its
                  average launch duration, measured with HIP events recorded around the
                  kernel on its own stream, against the 8 TB/s HBM peak
  cpu_baseline -- the unmodified reference (oracle/_ref/agrep, 1 core) on a bounded sample of
                  the same corpus, N=1 only
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PATTERN = b"approximatematch"
VARIANTS = (b"approximatematch", b"approximatematch", b"aproximatematch", b"approxXmatematch",
            b"approximatemmatcZ", b"apprximatemtch", b"appQoximRtematch")
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 12345


def cpu_baseline_extras(ref, host, sample_bytes, k, tmpdir):
    """the reference is single-threaded, so 'all host cores' = P independent
    processes on P record-aligned shards started together (aggregate = bytes / max wall time);
    plus the scalar C restatement (oracle) as a second, quirk-free CPU datapoint."""
    import _oracle as O
    out = {}
    procs = min(os.cpu_count() or 1, 32)
    pages = sample_bytes // 4096
    per = (pages + procs - 1) // procs
    paths = []
    try:
        for i in range(procs):
            lo, hi = i * per * 4096, min(sample_bytes, (i + 1) * per * 4096)
            if lo >= hi:
                break
            pth = os.path.join(tmpdir, "agh_bench_shard_%d_%d.txt" % (os.getpid(), i))
            host[lo:hi].tofile(pth)                       # pages end with a newline: record aligned
            paths.append(pth)
        cmd = [ref, "-V0", "-%d" % k, "-c", PATTERN.decode()]
        t0 = time.time()
        ps = [subprocess.Popen(cmd + [pth], stdout=subprocess.PIPE) for pth in paths]
        outs = [p_.communicate()[0] for p_ in ps]
        dt = time.time() - t0
        total = sum(int(o.split()[0]) for o in outs if o.strip())
        out["all_cores"] = {"value": round(sample_bytes / 1e9 / dt, 3), "unit": "GB/s", "cores": len(paths),
                            "kind": "reference", "seconds": round(dt, 3), "count": total,
                            "sample": "the same bytes as %d record-aligned shard files, one agrep process each, "
                                      "started together" % len(paths)}
    finally:
        for pth in paths:
            if os.path.exists(pth):
                os.unlink(pth)
    nb = min(sample_bytes, 256 << 20)
    t0 = time.time()
    cnt = O.asearch(PATTERN, k, host[:nb])[0]
    dt = time.time() - t0
    out["port"] = {"value": round(nb / 1e9 / dt, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                   "seconds": round(dt, 3), "count": int(cnt),
                   "sample": "first %.2f GiB, oracle/agrep_oracle.c orc_asearch (scalar restatement of asearch.c)" % (nb / 2**30)}
    return out


def cpu_baseline(text_dev, n_bytes, k, gpu_count_on_sample, sample_bytes):
    """Time the reference CPU agrep (1 core) on the first sample_bytes of the corpus."""
    ref = os.path.join(ROOT, "oracle", "_ref", "agrep")
    sample_bytes = min(sample_bytes, n_bytes)
    host = text_dev[:sample_bytes].cpu().numpy()
    if os.path.exists(ref):
        d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        path = os.path.join(d, "agh_bench_sample_%d.txt" % os.getpid())
        try:
            host.tofile(path)
            cmd = [ref, "-V0", "-%d" % k, "-c", PATTERN.decode(), path]
            subprocess.run(cmd, stdout=subprocess.PIPE)            # warm-up pass (page cache)
            t0 = time.time()
            out = subprocess.run(cmd, stdout=subprocess.PIPE).stdout
            dt = time.time() - t0
        finally:
            if os.path.exists(path):
                os.unlink(path)
        cnt = int(out.split()[0]) if out.strip() else -1
        extra = {}
        try:
            extra = cpu_baseline_extras(ref, host, sample_bytes, k, d)
        except Exception as e:                                  # never lose the headline over this
            extra = {"extras_error": str(e)[:200]}
        return {"value": round(sample_bytes / 1e9 / dt, 4), "unit": "GB/s", "cores": 1,
                **extra,
                "kind": "reference",
                "sample": "first %.2f GiB of the rank-0 shard, `agrep -V0 -%d -c %s` (sgrep.c:agrep() "
                          "path), page cache warm, 1 process" % (sample_bytes / 2**30, k, PATTERN.decode()),
                "seconds": round(dt, 3), "count": cnt,
                "count_equals_gpu": bool(cnt == gpu_count_on_sample)}
    import _oracle as O                                        # the restatement as a port
    sample_bytes = min(sample_bytes, 256 << 20)
    t0 = time.time()
    cnt = O.asearch(PATTERN, k, host[:sample_bytes])[0]
    dt = time.time() - t0
    return {"value": round(sample_bytes / 1e9 / dt, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "first %.2f GiB, oracle/agrep_oracle.c orc_asearch (scalar)" % (sample_bytes / 2**30),
            "seconds": round(dt, 3), "count": int(cnt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gib", type=float, default=4.0, help="corpus GiB per GPU")
    ap.add_argument("-k", type=int, default=2)
    ap.add_argument("--cpu-sample-gib", type=float, default=4.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import agrep_amd as A
    from agrep_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run)"
    # Test hooks (1-GPU boxes): AGH_BENCH_BACKEND=gloo + AGH_BENCH_ONE_GPU=1 run all ranks on
    # GPU 0 with the count reduction on CPU tensors -- exercises the multi-rank control flow.
    backend = os.environ.get("AGH_BENCH_BACKEND", "nccl")
    if os.environ.get("AGH_BENCH_ONE_GPU") == "1":
        local_rank = 0
    red_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    A.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    n_pages = int(args.gib * (1 << 30)) // 4096
    n = n_pages * 4096
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    first_page, my_pages = shard.shard_pages(n_pages * world, world, rank)
    assert my_pages == n_pages
    planted = A.corpus_fill_device(text.data_ptr(), n_pages, first_page=first_page, seed=SEED,
                                   variants=VARIANTS, plant_period=500)
    torch.cuda.synchronize()
    q = A.Query(PATTERN, args.k)
    info = q.info()
    agg = [0, 0]

    def step():
        # AGH_TIME_SWEEP: HIP events around k_sweep on the scan's stream, in every timed step
        res = q.scan_device(text.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP, time_scan=False)
        if world > 1:                                        # RCCL: the -c aggregate
            agg[0], agg[1] = shard.reduce_counts(res.n_matched, res.n_records, device=red_dev)
        return res

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    fence()
    t0 = time.perf_counter()
    sweep_ms = 0.0
    for _ in range(args.steps):
        res = step()
        sweep_ms += res.sweep_ms
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        matched_all = int(agg[0])
    else:
        matched_all = int(res.n_matched)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_bytes = n * world
        value = total_bytes / 1e9 / (elapsed / args.steps)
        sweep_avg_ms = sweep_ms / args.steps
        achieved = n / 1e6 / sweep_avg_ms                     # GB/s of the dominant kernel
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if abs(tj.get("bytes_per_launch_basis", 0) - n) < 1:
                    traffic = tj.get("hbm_read_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "GB/s scanned (k=2, m=16, -c count) + Mmatches/s",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: pattern 'approximatematch' (m=16), k=%d, "
                                   "%.2f GiB of newline-delimited records per GPU resident in HBM, "
                                   "count-only (-c)" % (args.k, args.gib),
                       "bytes_per_gpu": n,
                       "records_per_gpu": int(q.scan_device(text.data_ptr(), n).n_records),
                       "sharding": "disjoint page ranges per rank, RCCL all-reduce of the counts",
                       "engine": {1: "fullscan", 2: "q-gram sample filter + verify"}[res.engine],
                       "filter_sample": "q=%d bytes every h=%d bytes" % (info["filter_q"], info["filter_h"]),
                       "seed": SEED},
            "matched_records": matched_all,
            "mmatches_per_s": round(matched_all / 1e6 / (elapsed / args.steps), 3),
            "planted_records_rank0": int(sum(planted)),
            "candidates_per_step_rank0": int(res.n_candidates),
            "roofline": {"bound": "hbm", "kernel": "k_sweep<%d>" % info["filter_h"],
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": n,
                         "avg_launch_ms": round(sweep_avg_ms, 4)},
        }
        # secondary points of the metric: k = 0 and the streaming-read ceiling of this access pattern
        q0 = A.Query(PATTERN, 0)
        for _ in range(2):
            q0.scan_device(text.data_ptr(), n, flags=A.COUNT)
        t1 = time.perf_counter()
        for _ in range(5):
            r0 = q0.scan_device(text.data_ptr(), n, flags=A.COUNT)
        torch.cuda.synchronize()
        out["k0"] = {"value": round(n / 1e9 / ((time.perf_counter() - t1) / 5), 2), "unit": "GB/s",
                     "matched_records_rank0": int(r0.n_matched)}
        q0.close()
        A.probe_read_ms(text.data_ptr(), n)
        out["read_ceiling_gbps"] = round(n / 1e6 / min(A.probe_read_ms(text.data_ptr(), n) for _ in range(3)), 1)
        if world == 1 and not args.no_cpu_baseline:
            sb = int(args.cpu_sample_gib * (1 << 30)) // 4096 * 4096
            sb = min(sb, n)
            gpu_cnt = q.scan_device(text.data_ptr(), sb).n_matched
            out["cpu_baseline"] = cpu_baseline(text, n, args.k, int(gpu_cnt), sb)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    q.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
