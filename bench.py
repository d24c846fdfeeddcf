#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X agrep scanner.

Metric (BASELINE.json): GB/s scanned + Mmatches/s, k in {0, 2}, m = 16, 64 GiB synthetic
newline-delimited corpus at 1/2/4/8 GPUs.  STRONG scaling: the job is always the same 64 GiB
(16 shards x 4 GiB of SURVEY 8d's generator, seed 12345); rank r of N owns the contiguous
64/N GiB page range r (N = 8: 8 GiB per GPU = BASELINE configs[3]), resident in HBM before the
timed region starts.  A "step" is one complete `-c` scan of the rank's range through the C-ABI
(agh_scan_device: sweep + verify + count, one host sync); with N > 1 the RCCL all-reduce of the
counts is part of the same call (agh_scan_device_reduce: enqueued on the scan's stream, still one sync).

    python bench.py                     # 1 GPU, the whole 64 GiB
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with
  k0           -- the same K steps timed the same way with k = 0 (the metric's other point)
  roofline     -- dominant kernel (k_sweep_fused<H>: sweep + verify of a count-only scan in one
                  kernel; k_sweep<H> with AGH_FUSED=0): algorithmic bytes (1 B per corpus byte) / its
                  average launch duration, HIP events recorded around every launch on the
                  stream it is launched on, against the 8 TB/s HBM peak; `traffic` = HBM read
                  bytes per launch from rocprofv3's FETCH_SIZE counter, measured in THIS run by
                  a child process (or null)
  cpu_baseline -- the unmodified reference (oracle/_ref/agrep, 1 core and all cores) on a
                  bounded sample of the same corpus, N = 1 only
  c2_records / c3 / c5 -- the other BASELINE.json configs on resident text, N = 1 only, measured after
                  (outside) the headline's timed region with their own steps: C2 = m=16 k=2, 4 GiB, numbered
                  scan + device gather of the matched records (agh_scan_device_emit); C3 = m=48 k=3 -i,
                  16 GiB, count-only and numbered; C5 = -f 1024 patterns (8..12 B) k=1, 8 GiB (one GPU's
                  share of 32 GiB / 4), count-only.  Each block: value (GB/s), ms_per_step, matched vs
                  planted, and a roofline sub-block for the kernel that reads every byte (HIP events on the
                  scan's stream; C5 with its own in-run FETCH_SIZE pass)
"""
import argparse
import csv
import glob
import json
import os
import random
import shutil
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
VARIANT_EDITS = (0, 0, 1, 1, 2, 2, 2)
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 12345


def ref_all_cores(ref, host, k, tmpdir, procs, records=False, opts=(), pattern=None):
    """The reference is single-threaded: 'all host cores' = `procs` independent agrep processes on
    `procs` record-aligned pieces of `host` (pages end with a newline), started together.
    -> (sum of the counts, seconds from the first start to the last exit, processes[, sha256 of the lines]);
    records: run WITHOUT -c -- the reference prints the matched records (s_output(), sgrep.c:1274-1333); the
    count is then the number of printed lines and the digest covers the pieces' outputs in file order.
    opts / pattern: further options (-i: checksg.c:129 sends every -i query with errors to asearch(), asearch.c:32-572)
    and another pattern.  The pieces start at page boundaries and no record crosses a page, so no occurrence
    straddles byte 49152 of a piece (quirk Q1 of the asearch path's file mode)"""
    from concurrent.futures import ThreadPoolExecutor
    n = host.size
    pages = n // 4096
    per = (pages + procs - 1) // procs
    spans = [(i * per * 4096, min(n, (i + 1) * per * 4096)) for i in range(procs) if i * per * 4096 < n]
    paths = [os.path.join(tmpdir, "agh_bench_shard_%d_%d.txt" % (os.getpid(), i)) for i in range(len(spans))]
    try:
        with ThreadPoolExecutor(max_workers=8) as ex:             # (tofile releases the GIL)
            list(ex.map(lambda a: host[a[0][0]:a[0][1]].tofile(a[1]), zip(spans, paths)))
        cmd = [ref, "-V0"] + list(opts) + ["-%d" % k] + ([] if records else ["-c"]) + [(pattern or PATTERN).decode()]
        t0 = time.time()
        if records:
            # (stdout to files: 32 pipes read one after the other would stall the writers)
            outs_p = [pth + ".out" for pth in paths]
            fhs = [open(o, "wb") for o in outs_p]
            ps = [subprocess.Popen(cmd + [pth], stdout=fh) for pth, fh in zip(paths, fhs)]
            for p_ in ps:
                p_.wait()
            dt = time.time() - t0
            for fh in fhs:
                fh.close()
            import hashlib
            h = hashlib.sha256()
            lines = 0
            for o in outs_p:
                with open(o, "rb") as fh:
                    data = fh.read()
                h.update(data)
                lines += data.count(b"\n")
                os.unlink(o)
            return lines, dt, len(paths), h.hexdigest()
        ps = [subprocess.Popen(cmd + [pth], stdout=subprocess.PIPE) for pth in paths]
        outs = [p_.communicate()[0] for p_ in ps]
        dt = time.time() - t0
    finally:
        for pth in paths:
            if os.path.exists(pth):
                os.unlink(pth)
            if os.path.exists(pth + ".out"):
                os.unlink(pth + ".out")
    return sum(int(o.split()[0]) for o in outs if o.strip()), dt, len(paths)


def cpu_baseline_extras(ref, host, sample_bytes, k, tmpdir):
    """the all-cores leg on the sample, plus the scalar C restatement (oracle) as a
    second, quirk-free CPU datapoint."""
    import _oracle as O
    out = {}
    total, dt, np_ = ref_all_cores(ref, host[:sample_bytes], k, tmpdir, min(os.cpu_count() or 1, 32))
    out["all_cores"] = {"value": round(sample_bytes / 1e9 / dt, 3), "unit": "GB/s", "cores": np_,
                        "kind": "reference", "seconds": round(dt, 3), "count": total,
                        "sample": "the same bytes as %d record-aligned shard files, one agrep process each, "
                                  "started together" % np_}
    nb = min(sample_bytes, 256 << 20)
    t0 = time.time()
    cnt = O.asearch(PATTERN, k, host[:nb])[0]
    dt = time.time() - t0
    out["port"] = {"value": round(nb / 1e9 / dt, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                   "seconds": round(dt, 3), "count": int(cnt),
                   "sample": "first %.2f GiB, oracle/agrep_oracle.c orc_asearch (scalar restatement of asearch.c)" % (nb / 2**30)}
    return out


def reference_all_shards(text_dev, n_bytes, k, q, shard_bytes, first_shard_host, opts=(), pattern=None, make_shard=None,
                         what=None):
    """Parity of the WHOLE corpus against the reference CPU agrep -- the match SET, not only its size: shard by
    shard (4 GiB each: the SURVEY 8d shards) through /dev/shm, every shard cut into one file per core, one
    reference process per file printing the matched records; beside it the GPU's count of the same shard
    (count-only scan) and the sha256 of the records the GPU returns for it (agh_scan_device_emit with the
    delimiter behind every record: byte for byte what the reference prints).  Not a timing leg."""
    import hashlib
    import agrep_amd as A
    ref = os.path.join(ROOT, "oracle", "_ref", "agrep")
    d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    procs = min(os.cpu_count() or 1, 32)
    n_shards = n_bytes // shard_bytes
    t0 = time.time()
    ref_counts, gpu_counts, ref_sha, gpu_sha, gpu_recs = [], [], [], [], []
    ref_seconds = 0.0
    for sh in range(n_shards):
        lo = sh * shard_bytes
        if make_shard is not None:              # the shard is generated into text_dev[0:shard_bytes) now
            make_shard(sh)
            lo = 0
        host = first_shard_host if (sh == 0 and first_shard_host is not None and first_shard_host.size == shard_bytes) \
            else text_dev[lo:lo + shard_bytes].cpu().numpy()
        cnt, dt, _, digest = ref_all_cores(ref, host, k, d, procs, records=True, opts=opts, pattern=pattern)
        del host
        ref_counts.append(int(cnt))
        ref_sha.append(digest)
        ref_seconds += dt
        gpu_counts.append(int(q.scan_device(text_dev.data_ptr() + lo, shard_bytes, flags=A.COUNT,
                                            time_sweep=False, time_scan=False).n_matched))
        h = hashlib.sha256()
        _, batches = q.scan_device_emit(text_dev.data_ptr() + lo, shard_bytes, flags=A.EMIT_TAIL_DELIM,
                                        summarize=True, hasher=h)
        gpu_sha.append(h.hexdigest())
        gpu_recs.append(sum(b[0] for b in batches))
    all_ref, all_gpu = hashlib.sha256("".join(ref_sha).encode()), hashlib.sha256("".join(gpu_sha).encode())
    return {"shards": n_shards, "shard_bytes": shard_bytes, "processes_per_shard": procs,
            "reference_count": sum(ref_counts), "gpu_count": sum(gpu_counts), "gpu_records_returned": sum(gpu_recs),
            "shards_equal": sum(1 for a, b in zip(ref_counts, gpu_counts) if a == b),
            "records_sha256_shards_equal": sum(1 for a, b in zip(ref_sha, gpu_sha) if a == b),
            "records_sha256_equal": ref_sha == gpu_sha and gpu_recs == ref_counts,
            "records_sha256_of_shard_digests": {"reference": all_ref.hexdigest(), "gpu": all_gpu.hexdigest()},
            "equal": ref_counts == gpu_counts, "reference_seconds": round(ref_seconds, 2),
            "reference_GBps_all_cores": round(n_shards * shard_bytes / 1e9 / max(ref_seconds, 1e-9), 2),
            "wall_seconds": round(time.time() - t0, 1),
            "what": what or ("`agrep -V0 -%d %s` (unmodified reference, sgrep.c path, printing the matched records) over every "
                             "byte of the corpus; per shard: its line count against the GPU's -c count, the sha256 of its "
                             "lines against the sha256 of the records agh_scan_device_emit returns" % (k, PATTERN.decode()))}


def reference_all_shards_nocase(n_bytes, k, shard_bytes):
    """The asearch() path at full size (asearch.c:32-572: what EVERY -i query with errors runs, checksg.c:129): the
    same 16 shards of SURVEY 8d's generator with half of the letters upper-cased, `agrep -V0 -i -k pattern` (all
    cores, printing) against agh_scan_device_emit of a nocase query, sha256 per shard.  The shards are generated
    one after the other into a scratch buffer (the lower-case corpus stays resident next to it)."""
    import torch
    import agrep_amd as A
    buf = torch.empty(shard_bytes, dtype=torch.uint8, device="cuda")
    pages = shard_bytes // 4096

    def make_shard(sh):
        A.corpus_fill_device(buf.data_ptr(), pages, first_page=sh * pages, seed=SEED, variants=VARIANTS, plant_period=500,
                             upper_permille=500)
        torch.cuda.synchronize()
    try:
        with A.Query(PATTERN, k, nocase=True) as q:
            out = reference_all_shards(
                buf, n_bytes, k, q, shard_bytes, None, opts=("-i",), make_shard=make_shard,
                what="`agrep -V0 -i -%d %s` (unmodified reference: -i with errors = asearch(), asearch.c:32-572) over "
                     "every shard of the corpus with 50 %% of its letters upper-cased; per shard: its line count "
                     "against the GPU's -c count of the nocase query, the sha256 of its lines against the sha256 of "
                     "the records agh_scan_device_emit returns" % (k, PATTERN.decode()))
            out["engine"] = {1: "fullscan", 2: "q-gram sample filter + verify"}.get(
                int(q.scan_device(buf.data_ptr(), shard_bytes, flags=A.COUNT, time_sweep=False, time_scan=False).engine), "?")
    finally:
        del buf
        torch.cuda.empty_cache()
    return out


def cpu_baseline(text_dev, n_bytes, k, gpu_count_on_sample, sample_bytes, q=None, all_shards=False):
    """Time the reference CPU agrep (1 core) on the first sample_bytes of the corpus; with
    all_shards also run it (all cores) over the whole corpus and compare the counts."""
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
        if all_shards and q is not None and n_bytes >= 2 * sample_bytes and n_bytes % sample_bytes == 0:
            try:
                extra["all_shards"] = reference_all_shards(text_dev, n_bytes, k, q, sample_bytes, host)
                extra["all_shards_count_equals_gpu"] = bool(extra["all_shards"]["equal"])
                extra["all_shards_records_sha256_equal"] = bool(extra["all_shards"]["records_sha256_equal"])
            except Exception as e:
                extra["all_shards_error"] = str(e)[:200]
            try:
                extra["all_shards_nocase"] = reference_all_shards_nocase(n_bytes, k, sample_bytes)
                extra["all_shards_nocase_records_sha256_equal"] = bool(extra["all_shards_nocase"]["records_sha256_equal"])
            except Exception as e:
                extra["all_shards_nocase_error"] = str(e)[:200]
        return {"value": round(sample_bytes / 1e9 / dt, 4), "unit": "GB/s", "cores": 1,
                **extra,
                "kind": "reference",
                "sample": "shard 0 of 16 = the first %.2f GiB of the 64 GiB corpus, `agrep -V0 -%d -c %s` "
                          "(sgrep.c:agrep() path), page cache warm, 1 process"
                          % (sample_bytes / 2**30, k, PATTERN.decode()),
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


def measure_traffic(seg_gib, k, timeout_s, config="headline", kernels=("void k_sweep<", "void k_sweep_fused<")):
    """HBM read bytes of ONE launch of the dominant kernel, measured now: a child process runs a few scans of
    one segment of the same corpus under `rocprofv3 --pmc FETCH_SIZE` (its own pass, with
    --kernel-trace only, as MI355X_MICROARCH.md prescribes); FETCH_SIZE is in KiB and counts the
    128-byte requests of gfx950 as 64 bytes (the guide's correction: x 2).  None on any failure."""
    rp = shutil.which("rocprofv3")
    if not rp:
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="agh_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "AGH_BENCH_FORCE_DIST"):
            env.pop(v, None)
        cmd = [rp, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p",
               "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "--pmc-config", config,
               "--total-gib", str(seg_gib), "-k", str(k), "--steps", "3", "--warmup", "0"]
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=timeout_s)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, "rocprofv3 pass failed (rc %d)" % r.returncode
        vals = []
        for row in csv.DictReader(open(files[0])):
            if row.get("Counter_Name") == "FETCH_SIZE" and row.get("Kernel_Name", "").startswith(tuple(kernels)):
                vals.append(float(row["Counter_Value"]))
        if not vals:
            return None, "no %s rows in the counter file" % " / ".join(kernels)
        # one row per launch (the counter is summed over the XCDs by rocprofv3); drop nothing
        per_launch = sum(vals) / len(vals)
        return int(per_launch * 1024 * 2), "rocprofv3 --pmc FETCH_SIZE, %d launches of %g GiB, KiB x 1024 x 2" % (len(vals), seg_gib)
    except Exception as e:                                      # never lose the headline over this
        return None, "traffic pass failed: %s" % str(e)[:120]
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ---- the other BASELINE.json configs (N = 1, resident text, outside the headline's timed region) ----------
def c3_pattern_and_variants():
    """m = 48 pattern and its 0..4-edit variants (tests/test_gpu_fullsize.py::test_c3_*: the same recipe)"""
    import random
    rng = random.Random(48)
    pat = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(48))
    vs = [pat]
    for edits in (1, 2, 3, 4):
        v = bytearray(pat)
        for _ in range(edits):
            op, pos = rng.randint(0, 2), rng.randrange(4, len(v) - 4)
            if op == 0:
                v[pos] = ord("Q")
            elif op == 1:
                del v[pos]
            else:
                v.insert(pos, ord("Z"))
        vs.append(bytes(v))
    return pat, tuple(vs)


def c5_patterns_and_variants():
    """1024 patterns of 8..12 bytes and planted 0-, 1- and 3-edit variants (test_c5_1024_patterns_k1_8gib)"""
    import random
    rng = random.Random(1024)
    pats = set()
    while len(pats) < 1024:
        pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(8, 12))))
    pats = sorted(pats)
    rng = random.Random(7)

    def edit(p, edits):
        a = bytearray(p)
        for _ in range(edits):
            op, at = rng.randint(0, 2), rng.randrange(1, len(a) - 1)
            if op == 0:
                a[at] = ord("Q")
            elif op == 1:
                del a[at]
            else:
                a.insert(at, ord("Z"))
        return bytes(a)
    base = [pats[3], pats[500], pats[900]]
    variants = (base[0], base[1], edit(base[0], 1), edit(base[1], 1), edit(base[2], 1), edit(base[0], 3), edit(base[2], 3))
    return pats, variants


def c3_pinned_m29(A, torch, buf, n):
    import hashlib
    ref = os.path.join(ROOT, "oracle", "_ref", "agrep")
    if not os.path.exists(ref):
        return {"skipped": "oracle/_ref/agrep not built"}
    rng = random.Random(29)
    pat = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(29))
    vs = [pat]
    for edits in (1, 2, 3, 4):
        v = bytearray(pat)
        for _ in range(edits):
            op, pos = rng.randint(0, 2), rng.randrange(3, len(v) - 3)
            if op == 0:
                v[pos] = ord("Q")
            elif op == 1:
                del v[pos]
            else:
                v.insert(pos, ord("Z"))
        vs.append(bytes(v))
    planted = A.corpus_fill_device(buf.data_ptr(), n // 4096, seed=29, variants=tuple(vs), plant_period=500, upper_permille=500)
    want = int(sum(planted[:4]))
    d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    procs = min(os.cpu_count() or 1, 32)
    piece = min(n, 4 << 30)
    t0 = time.time()
    ref_sha, gpu_sha, ref_lines, gpu_recs, ref_s = [], [], 0, 0, 0.0
    with A.Query(pat, 3, nocase=True) as q:
        info = q.info()
        sec, r, sweep, launches, dev = timed_steps(torch, lambda: q.scan_device(buf.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP | A.TIME_SCAN), 5)
        for lo in range(0, n, piece):
            host = buf[lo:lo + piece].cpu().numpy()
            cnt, dt, _, digest = ref_all_cores(ref, host, 3, d, procs, records=True, opts=("-i",), pattern=pat)
            del host
            ref_lines += int(cnt)
            ref_s += dt
            ref_sha.append(digest)
            h = hashlib.sha256()
            _, batches = q.scan_device_emit(buf.data_ptr() + lo, piece, flags=A.EMIT_TAIL_DELIM, summarize=True, hasher=h)
            gpu_sha.append(h.hexdigest())
            gpu_recs += sum(b[0] for b in batches)
    return {"workload": "m=29 (the longest pattern maskgen.c:201-208 accepts) k=3 -i, %d GiB resident, 50 %% of the letters "
                        "upper-cased: `agrep -V0 -i -3 <pattern>` (asearch(), asearch.c:32-572; %d processes per %d GiB "
                        "piece, printing) against agh_scan_device_emit" % (n >> 30, procs, piece >> 30),
            "pattern": pat.decode(), "value": round(n / 1e9 / sec, 2), "unit": "GB/s", "ms_per_step": round(sec * 1e3, 4),
            "filter_sample": "q=%d bytes every h=%d bytes" % (info["filter_q"], info["filter_h"]),
            "matched_records": int(r.n_matched), "planted_records_0_3_edits": want,
            "reference_lines": ref_lines, "gpu_records_returned": gpu_recs,
            "records_sha256_pieces_equal": sum(1 for a, b in zip(ref_sha, gpu_sha) if a == b), "pieces": len(ref_sha),
            "records_sha256_equal": bool(ref_sha == gpu_sha and ref_lines == gpu_recs == int(r.n_matched)),
            "reference_seconds": round(ref_s, 2), "reference_GBps_all_cores": round(n / 1e9 / max(ref_s, 1e-9), 2),
            "wall_seconds": round(time.time() - t0, 1),
            "roofline": roofline_block("k_sweep_fused<H=%d>" % info["filter_h"] if r.fused_segments else "k_sweep<H=%d>" % info["filter_h"],
                                       n, 5, sweep, launches)}


def timed_steps(torch, fn, steps, warmup=2):
    """-> (seconds per step, last result, sum of sweep_ms, sweep launches, sum of device_ms)"""
    for _ in range(warmup):
        r = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sweep = dev = 0.0
    launches = 0
    for _ in range(steps):
        r = fn()
        sweep += r.sweep_ms
        dev += r.device_ms
        launches += r.sweep_launches
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, r, sweep, launches, dev


def roofline_block(kernel, n_bytes, steps, sweep_ms, launches):
    per_launch = n_bytes * steps / max(launches, 1)
    avg = sweep_ms / max(launches, 1)
    ach = per_launch / 1e6 / max(avg, 1e-9)
    return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None,
            "algorithmic_bytes_per_launch": int(per_launch), "avg_launch_ms": round(avg, 4), "launches_timed": int(launches)}


def config_blocks(A, torch, steps):
    out = {}
    buf = torch.empty(16 << 30, dtype=torch.uint8, device="cuda")
    F = A.TIME_SWEEP | A.TIME_SCAN

    # C2: m = 16, k = 2, 4 GiB, the match SET: numbered scan + device gather of the matched records
    n = 4 << 30
    planted = A.corpus_fill_device(buf.data_ptr(), n // 4096, seed=SEED, variants=VARIANTS, plant_period=500)
    want = sum(c for c, e in zip(planted, VARIANT_EDITS) if e <= 2)
    with A.Query(PATTERN, 2) as q:
        got = {}

        def step():
            # (summarize: the callback only counts -- no Python object per record inside the timed region)
            res, batches = q.scan_device_emit(buf.data_ptr(), n, flags=F, summarize=True)
            got["records"] = sum(b[0] for b in batches)
            got["bytes"] = sum(b[1] for b in batches)
            return res
        sec, r, sweep, launches, dev = timed_steps(torch, step, steps)
        info = q.info()
    out["c2_records"] = {
        "workload": "BASELINE configs[1]: m=16 k=2, 4 GiB resident, numbered scan + match bounds + device gather of the "
                    "matched records, records copied back (agh_scan_device_emit)",
        "value": round(n / 1e9 / sec, 2), "unit": "GB/s", "ms_per_step": round(sec * 1e3, 4), "steps": steps,
        "device_ms_scan_only": round(dev / steps, 4), "matched_records": int(r.n_matched), "records_returned": got["records"],
        "record_bytes_returned": got["bytes"], "planted_records": int(want),
        "matched_equals_planted": bool(r.n_matched == want == got["records"]), "segments": int(r.n_segments),
        "roofline": roofline_block("k_sweep<H=%d, census> (the kernel of the numbered pipeline that reads every byte)" % info["filter_h"],
                                   n, steps, sweep, launches)}

    # C3: m = 48, k = 3, -i, 16 GiB: count-only (one fused launch) and with record numbers
    n = 16 << 30
    pat, vs = c3_pattern_and_variants()
    planted = A.corpus_fill_device(buf.data_ptr(), n // 4096, seed=9, variants=vs, plant_period=500, upper_permille=500)
    want = int(sum(planted[:4]))
    nsteps = max(steps // 2, 3)
    with A.Query(pat, 3, nocase=True) as q:
        info = q.info()
        sec, r, sweep, launches, dev = timed_steps(torch, lambda: q.scan_device(buf.data_ptr(), n, flags=A.COUNT | F), steps)
        secn, rn, sweepn, launchesn, devn = timed_steps(torch, lambda: q.scan_device(buf.data_ptr(), n, flags=F), nsteps)
    out["c3"] = {
        "workload": "BASELINE configs[2]: m=48 k=3 -i, 16 GiB resident (64-bit state words; the reference rejects m > 29: "
                    "anchored on the planted 0..3-edit records)",
        "value": round(n / 1e9 / sec, 2), "unit": "GB/s", "ms_per_step": round(sec * 1e3, 4), "steps": steps,
        "matched_records": int(r.n_matched), "planted_records": want, "matched_equals_planted": bool(r.n_matched == want),
        "filter_sample": "q=%d bytes every h=%d bytes" % (info["filter_q"], info["filter_h"]),
        "fused_segments": int(r.fused_segments), "lean_reruns": int(r.lean_reruns),
        "roofline": roofline_block(("k_sweep_fused<H=%d> (64-bit words)" if r.fused_segments else "k_sweep<H=%d>") % info["filter_h"],
                                   n, steps, sweep, launches),
        "numbered": {"value": round(n / 1e9 / secn, 2), "unit": "GB/s", "ms_per_step": round(secn * 1e3, 4),
                     "steps": nsteps, "matched_records": int(rn.n_matched), "segments": int(rn.n_segments),
                     "matched_equals_planted": bool(rn.n_matched == want),
                     "roofline": roofline_block("k_sweep<H=%d, census>" % info["filter_h"], n, nsteps, sweepn, launchesn)}}

    # C3's pinned neighbour: m = 29 is the longest pattern the reference's maskgen accepts with a newline delimiter
    # (maskgen.c:201-208) -- k = 3, -i, the same 16 GiB recipe; the reference runs it (asearch(), 0.23 GB/s per core),
    # so the match SET is compared: sha256 of its lines against the records the device returns, 4 GiB at a time
    try:
        out["c3"]["pinned_m29"] = c3_pinned_m29(A, torch, buf, n)
    except Exception as e:                                      # never lose the line over this
        out["c3"]["pinned_m29"] = {"error": str(e)[:300]}

    # C5: -f, 1024 patterns of 8..12 bytes, k = 1, 8 GiB (one GPU's share of 32 GiB over 4), count-only
    n = 8 << 30
    pats, variants = c5_patterns_and_variants()
    planted = A.corpus_fill_device(buf.data_ptr(), n // 4096, seed=55, variants=variants, plant_period=500)
    want_le1 = int(sum(planted[:5]))
    q = A.Query.multi(pats, k=1)
    try:
        sec, r, sweep, launches, dev = timed_steps(torch, lambda: q.scan_device(buf.data_ptr(), n, flags=A.COUNT | F), steps)
        lean2 = q.scan_device(buf.data_ptr(), 2 << 30, flags=A.COUNT)
        numb2 = q.scan_device(buf.data_ptr(), 2 << 30, flags=A.COUNT | A.FORCE_NUMBERED)
    finally:
        q.close()
    one_pass = bool(r.fused_segments)
    out["c5"] = {
        "workload": "BASELINE configs[4] with pattern lengths 8..12 (SURVEY 8d says 4..12: that set is the c5_as_worded block): "
                    "-f 1024 patterns, k=1, 8 GiB resident = one GPU's share of 32 GiB / 4, count-only (-c / -l).  The "
                    "reference ignores -# with -f (compat.c:34-37): the predicate is the union of the single-pattern one",
        "value": round(n / 1e9 / sec, 2), "unit": "GB/s", "ms_per_step": round(sec * 1e3, 4), "steps": steps,
        "matched_records": int(r.n_matched), "planted_records_0_1_edits": want_le1,
        "matched_ge_planted": bool(r.n_matched >= want_le1 > 0),
        "count_only_equals_numbered_on_2gib": bool(lean2.n_matched == numb2.n_matched),
        "candidates_per_step": int(r.n_candidates), "segments": int(r.n_segments), "lean_reruns": int(r.lean_reruns),
        "roofline": roofline_block("k_mscan (one pass: pair-table probes, exact gram table, k=1 side check)" if one_pass
                                   else "k_sweep_multi (+ k_verify_multi)", n, steps, sweep, launches)}
    # C5 as SURVEY 8d words the set: 1024 patterns of 4..12 bytes, k = 1 -- pieces of two bytes, every position a
    # candidate, four records in five match: candidate / delimiter bits per tile, then a walk over the candidate bits with
    # exit at a record's first hit (agh_mtile.hip)
    n = 4 << 30
    rng = random.Random(1024)
    pw = set()
    while len(pw) < 1024:
        pw.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(4, 12))))
    pw = sorted(pw)
    A.corpus_fill_device(buf.data_ptr(), n // 4096, seed=5, variants=tuple(pw[:7]), plant_period=500)
    q = A.Query.multi(pw, k=1)
    try:
        sec, r, sweep, launches, dev = timed_steps(torch, lambda: q.scan_device(buf.data_ptr(), n, flags=A.COUNT | F), max(steps // 2, 3))
        lean2 = q.scan_device(buf.data_ptr(), 256 << 20, flags=A.COUNT)
        numb2 = q.scan_device(buf.data_ptr(), 256 << 20, flags=A.COUNT | A.FORCE_NUMBERED)
        # record-returning scans of the same set: record numbers from the census, matched records marked in the bitmap
        secn, rn, _, _, devn = timed_steps(torch, lambda: q.scan_device(buf.data_ptr(), 1 << 30, flags=A.COUNT | A.FORCE_NUMBERED | F), 3, warmup=1)
    finally:
        q.close()
    nsw = max(steps // 2, 3)
    out["c5_as_worded"] = {
        "workload": "SURVEY 8d's wording of configs[4]: -f 1024 patterns of 4..12 bytes, k=1, 4 GiB resident, count-only "
                    "(-c / -l); unpinned: the reference ignores -# with -f, the oracle is the union of 1024 single-pattern "
                    "scans (tests/test_gpu_fullsize.py: 64 MiB slice)",
        "value": round(n / 1e9 / sec, 2), "unit": "GB/s", "ms_per_step": round(sec * 1e3, 4), "steps": nsw,
        "matched_records": int(r.n_matched), "one_pass": bool(r.fused_segments), "lean_reruns": int(r.lean_reruns),
        "candidates_examined_per_step": int(r.n_candidates),
        "count_only_equals_numbered_on_256mib": bool(lean2.n_matched == numb2.n_matched),
        "numbered_1gib": {"value": round((1 << 30) / 1e9 / secn, 2), "unit": "GB/s", "ms_per_step": round(secn * 1e3, 4),
                          "device_ms": round(devn / 3, 4), "matched_records": int(rn.n_matched),
                          "what": "census + k_mtile marking record numbers + bitmap count (round 5: k_dense_multi, 28 GB/s)"},
        "roofline": roofline_block("k_mtile (candidate bits per 4 KiB tile + walk with exit at a record's first hit)" if r.fused_segments else "k_dense_multi",
                                   n, nsw, sweep, launches)}
    del buf
    torch.cuda.empty_cache()
    return out


def c5_file_hits_job(A, torch, dist, comm, rank, world, backend, fence, args, n_files=32, steps=5):
    """BASELINE configs[4] at N > 1: 32 files x 1 GiB of the C5 corpus, 1024 patterns of 8..12 bytes, k = 1, dealt in
    blocks to G = min(N, 4) GPUs (rank r < G holds files [r * 32 / G, (r + 1) * 32 / G) resident in HBM).  A step = the
    rank's files scanned one by one + ONE agh_reduce_file_hits over all ranks of the C-ABI's communicator.  Timed twice:
    `every_byte` -- count-only scans (hit = count > 0; the throughput figure: every byte is read) -- and `dash_l` -- the
    -l flag itself (asearch.c:130-161 stops at a file's first match).  The vector is cross-checked against
    torch.distributed's own MAX all-reduce of the same local flags."""
    G = min(world, 4)
    per = n_files // G
    fbytes = (args.c5_file_mib << 20) // 4096 * 4096
    pats, variants = c5_patterns_and_variants()
    mine = list(range(rank * per, (rank + 1) * per)) if rank < G else []
    pages = fbytes // 4096
    planted_files = []
    buf = q = None
    ok, why = 1, ""
    try:                                        # everything a rank does alone: no collective in here
        buf = torch.empty(max(len(mine), 1) * fbytes, dtype=torch.uint8, device="cuda")
        for i, f in enumerate(mine):
            # (one file in four holds planted patterns; with one error allowed the other files have chance
            # matches of their own -- about 4 per MiB -- so on this corpus -l lists every file of a GiB)
            pl = A.corpus_fill_device(buf.data_ptr() + i * fbytes, pages, first_page=f * pages, seed=55, variants=variants,
                                      plant_period=500 if f % 4 == 0 else 1 << 30)
            planted_files.append(int(sum(pl[:5])))
        torch.cuda.synchronize()
        q = A.Query.multi(pats, k=1)
        for i, f in enumerate(mine[:1]):        # (one scan of each kind before anybody waits in a collective)
            q.scan_device(buf.data_ptr(), fbytes, flags=A.COUNT, time_sweep=False, time_scan=False)
            q.scan_device(buf.data_ptr(), fbytes, flags=A.FILENAMEONLY, time_sweep=False, time_scan=False)
    except Exception as e:
        ok, why = 0, str(e)[:200]
    # the ranks agree to run the job or to skip it TOGETHER: a rank that failed above must not leave the others waiting
    # in the collectives below
    flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if q is not None:
            q.close()
        del buf
        torch.cuda.empty_cache()
        return {"skipped": "a rank could not set the job up" + (": " + why if why else "")}
    res = {}
    try:
        for name, fl in (("every_byte", A.COUNT), ("dash_l", A.FILENAMEONLY)):
            def step():
                hits = [0] * n_files
                cnt = 0
                for i, f in enumerate(mine):
                    r = q.scan_device(buf.data_ptr() + i * fbytes, fbytes, flags=fl, time_sweep=False, time_scan=False)
                    hits[f] = 1 if r.n_matched else 0
                    cnt += int(r.n_matched)
                if comm is not None:
                    return comm.reduce_file_hits(hits), hits, cnt
                from agrep_amd import shard
                return shard.reduce_file_hits(hits, device="cpu"), hits, cnt
            step()
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                vec, local, cnt = step()
            fence()
            el = time.perf_counter() - t0
            t = torch.tensor([el, float(cnt)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            allr = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allr, t)
            chk = torch.tensor(local, dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(chk, op=dist.ReduceOp.MAX)
            res[name] = {"ms_per_step": round(max(float(a[0]) for a in allr) / steps * 1e3, 4),
                         "rank_ms_per_step": [round(float(a[0]) / steps * 1e3, 4) for a in allr],
                         "files_listed": int(sum(1 for v in vec if v)),
                         "vector_equals_torch_all_reduce": bool([bool(v) for v in vec] == [bool(x) for x in chk.tolist()]),
                         "matched_records_all_ranks": int(sum(float(a[1]) for a in allr)) if name == "every_byte" else None}
    finally:
        q.close()
        del buf
        torch.cuda.empty_cache()
    ci = comm.info() if comm is not None else {"nranks": 0}
    tot = n_files * fbytes
    return {"workload": "BASELINE configs[4]: -f 1024 patterns (8..12 B) k=1, %d files x %d MiB resident on %d of %d GPU(s), "
                        "per step every file scanned + one agh_reduce_file_hits (ncclAllReduce max) over all %d ranks"
                        % (n_files, fbytes >> 20, G, world, world),
            "value": round(tot / 1e9 / (res["every_byte"]["ms_per_step"] / 1e3), 2), "unit": "GB/s", "files": n_files,
            "gpus_with_files": G, "rccl_ranks": int(ci["nranks"]), "steps": steps,
            "hit_reduction": ("agh_reduce_file_hits (RCCL inside the C-ABI)" if backend == "nccl" else
                              "agh_reduce_file_hits over agh_comm_init_custom (torch.distributed/%s: test hook)" % backend)
                             if comm is not None else "torch.distributed/" + backend,
            "planted_files_of_rank0": int(sum(1 for c in planted_files if c)) if rank == 0 else None,
            **res}


def engine_blocks(A, torch, steps=5):
    """The engines BEHIND the sample filter, 4 GiB resident each, count-only, every one with a roofline sub-block
    (algorithmic bytes = the text, time = the scan's device time between HIP events on its stream: the engine's
    kernel, its replay kernel and the count) and its count on the first 64 MiB against the oracle's
    (oracle/agrep_oracle.c: orc_asearch on the tables agh_compile_pattern makes, orc_asearch_costs, orc_wm_count
    for m > 29).  `matching` is the word of the reference's own timing table (agrep.ps.2 table 3: BASELINE.md)."""
    import _oracle as O
    n = 4 << 30
    sl = 64 << 20
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    F = A.TIME_SWEEP | A.TIME_SCAN
    ENG = {1: "fullscan / table engine", 2: "q-gram sample filter + verify"}
    rng = random.Random(40)
    pat40 = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(40))

    def tables(pat, delim):
        tb = A.compile_pattern(pat, delim=delim)
        return O.tables_from_golden({"Mask": list(tb.Mask), "Init0": tb.Init0, "Init1": tb.Init1, "NO_ERR_MASK": tb.NO_ERR_MASK,
                                     "endposition": tb.endposition, "D_endpos": tb.D_endpos, "wildmask": tb.wildmask,
                                     "AND": tb.AND}, tb.M, dlen=len(delim))

    cases = []      # (name, corpus, query factory, oracle(slice bytes) -> count, flags, slice bytes)
    for k in (1, 2, 3):
        cases.append(("matching_k%d" % k, "nl", lambda k=k: A.Query(b"matching", k),
                      lambda t, k=k: O.asearch(b"matching", k, t)[0], 0, sl))
    for k in (0, 1, 2):
        cases.append(("approx#match_k%d" % k, "nl", lambda k=k: A.Query.pattern(b"approx#match", k),
                      lambda t, k=k: O.asearch_tables(tables(b"approx#match", b"\n"), k, t)[0], 0, sl))
    for k in (1, 2):
        cases.append(("approx#match_I2_k%d" % k, "nl", lambda k=k: A.Query.pattern(b"approx#match", k).set_costs(2, 1, 1),
                      lambda t, k=k: O.asearch_tables_costs(tables(b"approx#match", b"\n"), k, (2, 1, 1), t)[0], 0, sl))
    for k in (0, 1, 2):
        cases.append(("approx#match_d_e_space_k%d" % k, "nl", lambda k=k: A.Query.pattern(b"approx#match", k, delim=b"e "),
                      lambda t, k=k: O.asearch_tables(tables(b"approx#match", b"e "), k, t, delim=b"e ")[0], 0, sl))
    for k in (0, 1):
        cases.append(("approx#match_1700B_records_k%d" % k, "nl", lambda k=k: A.Query.pattern(b"approx#match", k, delim=b"s\n"),
                      lambda t, k=k: O.asearch_tables(tables(b"approx#match", b"s\n"), k, t, delim=b"s\n")[0], 0, sl))
    cases.append(("m40_nocase_k6_filter", "upper", lambda: A.Query(pat40, 6, nocase=True),
                  lambda t: O.wm_count(pat40, 6, t, nocase=True)[0], 0, 16 << 20))
    cases.append(("m40_nocase_k6_fullscan_64bit_words", "upper", lambda: A.Query(pat40, 6, nocase=True),
                  lambda t: O.wm_count(pat40, 6, t, nocase=True)[0], A.FORCE_FULLSCAN, 16 << 20))

    out = {"workload": "engines behind the sample filter: 4 GiB of the SURVEY 8d corpus resident, count-only (-c), %d timed "
                       "steps each; slice = the first 64 MiB (16 MiB for m = 40) against the oracle" % steps}
    have = None
    host_cache = {}
    for name, corpus, make, oracle, flags, slb in cases:
        try:
            if have != corpus:
                A.corpus_fill_device(buf.data_ptr(), n // 4096, seed=SEED, variants=VARIANTS + (pat40,), plant_period=500,
                                     upper_permille=500 if corpus == "upper" else 0)
                torch.cuda.synchronize()
                have = corpus
                host_cache.clear()
            q = make()
            try:
                sec, r, sweep, launches, dev = timed_steps(torch, lambda: q.scan_device(buf.data_ptr(), n, flags=A.COUNT | F | flags), steps, warmup=1)
                got = int(q.scan_device(buf.data_ptr(), slb, flags=A.COUNT | flags, time_sweep=False, time_scan=False).n_matched)
            finally:
                q.close()
            if slb not in host_cache:
                host_cache[slb] = buf[:slb].cpu().numpy()
            t0 = time.time()
            want = int(oracle(host_cache[slb]))
            # (count-only scans on the sample filter report their one fused kernel as sweep_ms, not device_ms)
            dev_ms = (dev if dev > 0 else sweep) / steps
            ach = n / 1e6 / max(dev_ms, 1e-9)
            out[name] = {"value": round(n / 1e9 / sec, 1), "unit": "GB/s", "ms_per_step": round(sec * 1e3, 4),
                         "engine": ENG.get(int(r.engine), str(int(r.engine))), "matched_records": int(r.n_matched),
                         "slice_count": got, "slice_oracle_count": want, "slice_equals_oracle": bool(got == want),
                         "oracle_seconds": round(time.time() - t0, 2),
                         "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                      "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "device_ms": round(dev_ms, 4),
                                      "algorithmic_bytes_per_launch": n}}
        except Exception as e:                                  # never lose the line over this
            out[name] = {"error": str(e)[:200]}
    del buf
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # ~1.1 s timed per k at 64 GiB on one GPU
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--total-gib", type=float, default=64.0,
                    help="corpus GiB of the whole job (BASELINE: 64), split evenly over the ranks")
    ap.add_argument("-k", type=int, default=2)
    ap.add_argument("--cpu-sample-gib", type=float, default=4.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-all-shards", action="store_true",
                    help="skip the reference run over every shard of the corpus (N = 1 only, ~1 min)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 FETCH_SIZE pass")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-config", default="headline", help=argparse.SUPPRESS)
    ap.add_argument("--no-configs", action="store_true", help="skip the c2_records / c3 / c5 blocks (N = 1 only, ~30 s)")
    ap.add_argument("--config-steps", type=int, default=20)
    ap.add_argument("--no-engines", action="store_true", help="skip the engines block (N = 1 only, ~20 s)")
    ap.add_argument("--no-c5-files", action="store_true", help="skip the configs[4] file job with the -l hit vector (N > 1 only)")
    ap.add_argument("--c5-file-mib", type=int, default=1024, help="size of one of the 32 files of the configs[4] job")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per
        # GPU over RCCL, the contract's own command line) and become that launcher
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    import torch
    import torch.distributed as dist
    import agrep_amd as A
    from agrep_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run)"
    # Test hooks (1-GPU boxes): AGH_BENCH_BACKEND=gloo + AGH_BENCH_ONE_GPU=1 run all ranks on
    # GPU 0 with the count reduction through torch/gloo (RCCL refuses two ranks on one device).
    backend = os.environ.get("AGH_BENCH_BACKEND", "nccl")
    if os.environ.get("AGH_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    A.set_device(local_rank)
    comm = None
    # AGH_BENCH_FORCE_DIST=1 (test hook): take the multi-rank code path -- process group, the C-ABI's
    # own RCCL communicator, the all-reduce in every step -- even with a single rank
    dist_on = world > 1 or os.environ.get("AGH_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # the C-ABI's own RCCL communicator: rank 0's unique id travels over torch.distributed
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(A.Comm.unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, src=0)
            comm = A.Comm(bytes(uid.cpu().numpy().tobytes()), world, rank)
        else:
            # (test hook, one-GPU boxes: RCCL refuses two ranks on one device) the SAME C-ABI calls --
            # agh_scan_device_reduce, agh_reduce_file_hits -- on a communicator over the caller's transport
            # (agh_comm_init_custom), here torch.distributed/gloo
            dist.init_process_group(backend)

            def _allreduce(vals, elem):
                t = torch.tensor(vals, dtype=torch.int64)
                dist.all_reduce(t, op=dist.ReduceOp.SUM if elem == 8 else dist.ReduceOp.MAX)
                return t.tolist()
            comm = A.Comm.custom(_allreduce, world, rank)

    if args.pmc_child and args.pmc_config == "c5":       # under rocprofv3 --pmc: the C5 scan, nothing else
        n5 = int(args.total_gib * (1 << 30)) // 4096 * 4096
        t5 = torch.empty(n5, dtype=torch.uint8, device="cuda")
        pats, variants = c5_patterns_and_variants()
        A.corpus_fill_device(t5.data_ptr(), n5 // 4096, seed=55, variants=variants, plant_period=500)
        with A.Query.multi(pats, k=1) as q5:
            for _ in range(args.steps):
                q5.scan_device(t5.data_ptr(), n5, flags=A.COUNT, time_sweep=False, time_scan=False)
        torch.cuda.synchronize()
        return

    total_pages = int(args.total_gib * (1 << 30)) // 4096
    first_page, n_pages = shard.shard_pages(total_pages, world, rank)
    n = n_pages * 4096
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    planted = A.corpus_fill_device(text.data_ptr(), n_pages, first_page=first_page, seed=SEED,
                                   variants=VARIANTS, plant_period=500)
    torch.cuda.synchronize()

    if args.pmc_child:                          # under rocprofv3 --pmc: a few scans, nothing else
        with A.Query(PATTERN, args.k) as qc:
            for _ in range(args.steps):
                qc.scan_device(text.data_ptr(), n, flags=A.COUNT, time_sweep=False, time_scan=False)
        torch.cuda.synchronize()
        return

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(q):
        """W untimed + K timed steps; -> (seconds max over ranks, last result, sweep ms, sweep launches,
        matched records of the whole job)"""
        agg = [0, 0]

        def step():
            # AGH_TIME_SWEEP: HIP events around every k_sweep launch on the scan's stream
            if dist_on and comm is not None:
                # scan + the -c aggregate in one call: the counts stay on the device, ncclAllReduce is queued on
                # the scan's stream behind the kernels, one host synchronisation per step (agh_scan_device_reduce)
                res, (agg[0], agg[1]) = q.scan_device_reduce(comm, text.data_ptr(), n, flags=A.COUNT, time_sweep=True)
                return res
            res = q.scan_device(text.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP, time_scan=False)
            if dist_on:
                agg[0], agg[1] = shard.reduce_counts(res.n_matched, res.n_records, device="cpu")
            return res

        for _ in range(args.warmup):
            res = step()
        fence()
        t0 = time.perf_counter()
        sweep_ms, launches = 0.0, 0
        for _ in range(args.steps):
            res = step()
            sweep_ms += res.sweep_ms
            launches += res.sweep_launches
        fence()
        elapsed = time.perf_counter() - t0
        per_rank = None
        if dist_on:
            # every rank's own numbers travel to rank 0: its time, its sweep launches, and what the C-ABI's RCCL
            # communicator says about itself (agh_comm_info: a SCALE record can show that RCCL saw N ranks)
            ci = comm.info() if comm is not None else {"rank": rank, "nranks": 0, "device": local_rank}
            mine = torch.tensor([elapsed, sweep_ms, launches, n, ci["nranks"], ci["rank"], ci["device"], res.n_matched],
                                dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [[float(x) for x in t.tolist()] for t in allr]
            elapsed = max(r[0] for r in per_rank)
            matched_all = int(agg[0])
        else:
            matched_all = int(res.n_matched)
        return elapsed, res, sweep_ms, launches, matched_all, per_rank

    q = A.Query(PATTERN, args.k)
    info = q.info()
    elapsed, res, sweep_ms, launches, matched_all, per_rank = timed_loop(q)
    q0 = A.Query(PATTERN, 0)
    info0 = q0.info()
    elapsed0, res0, sweep_ms0, launches0, matched0, per_rank0 = timed_loop(q0)

    # N > 1: BASELINE configs[4] as a job of its own -- 32 "files" of 1 GiB dealt to min(N, 4) GPUs, -f 1024 patterns
    # k = 1, and the -l hit vector through agh_reduce_file_hits (ncclAllReduce(max) inside the C-ABI; every rank of the
    # communicator takes part, the ones without files contribute zeros)
    c5_files = None
    if dist_on and not args.no_c5_files:
        try:
            c5_files = c5_file_hits_job(A, torch, dist, comm, rank, world, backend, fence, args)
        except Exception as e:                                  # never lose the headline over this
            c5_files = {"error": str(e)[:300]}

    # planted records of the whole job (all ranks), by number of edits
    pl = torch.tensor([int(x) for x in planted], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
    if dist_on:
        dist.all_reduce(pl, op=dist.ReduceOp.SUM)
    pl = [int(x) for x in pl.tolist()]
    planted_le = {kk: sum(c for c, e in zip(pl, VARIANT_EDITS) if e <= kk) for kk in (0, 1, 2)}

    if rank == 0:
        step_s = elapsed / args.steps
        total_bytes = total_pages * 4096
        value = total_bytes / 1e9 / step_s
        per_launch_bytes = n * args.steps / max(launches, 1)
        sweep_avg_ms = sweep_ms / max(launches, 1)
        achieved = per_launch_bytes / 1e6 / sweep_avg_ms          # GB/s of the dominant kernel
        out = {
            "metric": "GB/s scanned + Mmatches/s, k in {0,2}, m=16, 64 GiB corpus (value: k=%d, -c count)" % args.k,
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "BASELINE metric config (configs[3] corpus): pattern 'approximatematch' (m=16), "
                                   "k=%d, %.0f GiB of newline-delimited records in total = 16 shards x 4 GiB of "
                                   "the SURVEY 8d generator, split evenly over %d GPU(s), resident in HBM, "
                                   "count-only (-c)" % (args.k, args.total_gib, world),
                       "total_bytes": total_bytes, "bytes_per_gpu": n,
                       "segments_per_gpu": int(res.n_segments),
                       "sharding": "contiguous page range per rank; the only exchange is the RCCL all-reduce of "
                                   "the counts (agh_reduce_counts)" if world > 1 else "one GPU holds the whole corpus",
                       "count_reduction": (("agh_scan_device_reduce (RCCL ncclAllReduce inside the C-ABI, enqueued on the scan's "
                                            "stream with the counts in device memory: one host sync per step)" if backend == "nccl"
                                            else "agh_scan_device_reduce over agh_comm_init_custom (torch.distributed/%s: test hook)" % backend)
                                           if comm is not None else "none (one rank)"),
                       "engine": {1: "fullscan", 2: "q-gram sample filter + verify"}[res.engine],
                       "filter_sample": "q=%d bytes every h=%d bytes" % (info["filter_q"], info["filter_h"]),
                       "seed": SEED},
            "matched_records": matched_all,
            "planted_records": planted_le[min(args.k, 2)],
            "matched_equals_planted": bool(matched_all == planted_le[min(args.k, 2)]),
            "mmatches_per_s": round(matched_all / 1e6 / step_s, 3),
            "timed_region_s": round(elapsed, 4),
            "lean_reruns_last_step": int(res.lean_reruns),
            "candidates_per_step_rank0": int(res.n_candidates),
            "k0": {"value": round(total_bytes / 1e9 / (elapsed0 / args.steps), 2), "unit": "GB/s",
                   "ms_per_step": round(elapsed0 / args.steps * 1e3, 4), "steps": args.steps,
                   "matched_records": matched0, "planted_records": planted_le[0],
                   "mmatches_per_s": round(matched0 / 1e6 / (elapsed0 / args.steps), 3),
                   "filter_sample": "q=%d bytes every h=%d bytes" % (info0["filter_q"], info0["filter_h"]),
                   "sweep_avg_launch_ms": round(sweep_ms0 / max(launches0, 1), 4)},
            "roofline": {"bound": "hbm",
                         "kernel": ("k_sweep_fused<H=%d> (sweep + verify of a count-only scan in one kernel)"
                                    if res.fused_segments else "k_sweep<%d>") % info["filter_h"],
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": int(per_launch_bytes),
                         "avg_launch_ms": round(sweep_avg_ms, 4), "launches_timed": int(launches),
                         "whole_scan_frac": round(value / world / HBM_PEAK_GBPS, 4)},
        }
        if c5_files is not None:
            out["c5_files"] = c5_files
        if per_rank is not None:
            def rank_rows(rows):
                out_rows = []
                for r_ in rows:
                    el, sw, la, nb, cn, cr, cd, mt = r_
                    avg = sw / max(la, 1)
                    out_rows.append({"rank": int(cr), "device": int(cd), "rccl_ranks": int(cn), "bytes": int(nb),
                                     "ms_per_step": round(el / args.steps * 1e3, 4), "matched_records": int(mt),
                                     "roofline": {"bound": "hbm", "achieved": round(nb * args.steps / max(la, 1) / 1e6 / max(avg, 1e-9), 1),
                                                  "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                                  "frac": round(nb * args.steps / max(la, 1) / 1e6 / max(avg, 1e-9) / HBM_PEAK_GBPS, 4),
                                                  "avg_launch_ms": round(avg, 4), "launches_timed": int(la)}})
                return out_rows
            out["ranks"] = rank_rows(per_rank)
            out["rank_ms_per_step_max"] = max(r_["ms_per_step"] for r_ in out["ranks"])
            out["rank_ms_per_step_min"] = min(r_["ms_per_step"] for r_ in out["ranks"])
            out["rccl_ranks"] = int(per_rank[0][4])
            out["k0"]["ranks"] = [{"rank": r_["rank"], "ms_per_step": r_["ms_per_step"], "frac": r_["roofline"]["frac"]}
                                  for r_ in rank_rows(per_rank0)]
        A.probe_read_ms(text.data_ptr(), min(n, 8 << 30))
        rn = min(n, 8 << 30)
        out["read_ceiling_gbps"] = round(rn / 1e6 / min(A.probe_read_ms(text.data_ptr(), rn) for _ in range(3)), 1)
        if world == 1 and not args.no_cpu_baseline:
            sb = int(args.cpu_sample_gib * (1 << 30)) // 4096 * 4096
            sb = min(sb, n)
            gpu_cnt = q.scan_device(text.data_ptr(), sb, flags=A.COUNT).n_matched
            out["cpu_baseline"] = cpu_baseline(text, n, args.k, int(gpu_cnt), sb, q=q,
                                               all_shards=not args.no_all_shards)
        else:
            out["cpu_baseline"] = None
    q.close()
    q0.close()
    if rank == 0:
        if world == 1 and not args.no_configs:
            try:
                out.update(config_blocks(A, torch, args.config_steps))
            except Exception as e:                              # never lose the headline over this
                out["configs_error"] = str(e)[:300]
        if world == 1 and not args.no_engines:
            try:
                out["engines"] = engine_blocks(A, torch)
            except Exception as e:                              # never lose the headline over this
                out["engines_error"] = str(e)[:300]
        if world == 1 and not args.no_traffic:
            del text
            torch.cuda.empty_cache()
            seg_gib = per_launch_bytes / 2**30
            tr, note = measure_traffic(seg_gib, args.k, 240)
            out["roofline"]["traffic"] = tr
            out["roofline"]["traffic_source"] = note
            if tr:
                out["roofline"]["traffic_over_algorithmic"] = round(tr / per_launch_bytes, 4)
            if "c5" in out and "roofline" in out["c5"]:
                r5 = out["c5"]["roofline"]
                g5 = r5["algorithmic_bytes_per_launch"] / 2**30
                tr5, note5 = measure_traffic(g5, 1, 240, config="c5", kernels=("void k_mscan<", "void k_sweep_multi<"))
                r5["traffic"] = tr5
                r5["traffic_source"] = note5
                if tr5:
                    r5["traffic_over_algorithmic"] = round(tr5 / r5["algorithmic_bytes_per_launch"], 4)
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
