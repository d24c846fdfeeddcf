"""File -> answer through the streaming pipeline (agh_stage.cpp): a page-cache-warm file in /dev/shm,
in-process (HIP runtime up, query built): count-only, records through agh_scan_fd_emit, records through
the match-array path (whole input staged), by reader threads and device segment size; then the C CLI and
the reference CLI end to end.  PCIe-inclusive numbers: never the bench `value`.
usage: scripts/file_stream_r4.py [GiB, default 4]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(gib * (1 << 30)) // 4096 * 4096
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
path = "/dev/shm/agh_stream_r4.txt"
t.cpu().numpy().tofile(path)
del t
torch.cuda.empty_cache()
print("host cores:", os.cpu_count(), flush=True)


def best_of(fn, reps=4):
    xs = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        xs.append(time.perf_counter() - t0)
    return min(xs), r


try:
    for readers, seg in ((16, 1024), (32, 1024), (16, 256), (32, 256), (64, 256), (8, 1024)):
        os.environ["AGH_READERS"] = str(readers)
        os.environ["AGH_STREAM_SEG_MB"] = str(seg)
        q = A.Query(b"approximatematch", 2)
        fd = os.open(path, os.O_RDONLY)

        def count():
            os.lseek(fd, 0, os.SEEK_SET)
            return q.scan_fd(fd, flags=A.COUNT)[0].n_matched

        def records():
            os.lseek(fd, 0, os.SEEK_SET)
            res, b = q.scan_fd_emit(fd, summarize=True)
            return (sum(x[0] for x in b), sum(x[1] for x in b), len(b))

        def array():
            os.lseek(fd, 0, os.SEEK_SET)
            res, ms = q.scan_fd(fd, cap=200000)
            recs = q.fetch_records(ms)
            return len(recs)
        count(); records()
        tc, c = best_of(count)
        tr, r = best_of(records)
        print("readers %2d segment %4d MiB: count-only %.3f s (%.1f GB/s) -> %d | records (emit) %.3f s (%.1f GB/s) -> %d records, %d bytes, %d batches"
              % (readers, seg, tc, n / 1e9 / tc, c, tr, n / 1e9 / tr, r[0], r[1], r[2]), flush=True)
        if readers == 16 and seg == 1024:
            ta, a = best_of(array, 2)
            print("   match-array path (whole input staged, fetch_records): %.3f s (%.1f GB/s) -> %d records" % (ta, n / 1e9 / ta, a), flush=True)
        os.close(fd)
        q.close()
    for k in ("AGH_READERS", "AGH_STREAM_SEG_MB"):
        os.environ.pop(k, None)
    cli = os.path.join(ROOT, "agrep_amd", "agrep-hip")
    ref = os.path.join(ROOT, "oracle", "_ref", "agrep")
    for name, exe in (("agrep-hip", cli), ("reference", ref)):
        if not os.path.exists(exe):
            continue
        for args in (["-V0", "-2", "-c"], ["-V0", "-2"]):
            xs = []
            for rep in range(2 if name == "reference" else 3):
                t0 = time.time()
                out = subprocess.run([exe] + args + ["approximatematch", path], stdout=subprocess.PIPE).stdout
                xs.append(time.time() - t0)
            tag = out.split()[0].decode() if "-c" in args else "%d lines" % out.count(b"\n")
            print("%-10s %-12s %.3f s  %.2f GB/s  -> %s" % (name, " ".join(args), min(xs), n / 1e9 / min(xs), tag), flush=True)
finally:
    os.unlink(path)
