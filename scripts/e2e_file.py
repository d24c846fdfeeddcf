"""End-to-end file -> answer: the C CLI (read + H2D staging + scan) vs the reference CLI on a
page-cache-warm file in /dev/shm.  Reported separately from the HBM-resident number."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(gib * (1 << 30)) // 4096 * 4096
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
path = "/dev/shm/agh_e2e.txt"
t.cpu().numpy().tofile(path)
del t
torch.cuda.empty_cache()
cli = os.path.join(ROOT, "agrep_amd", "agrep-hip")
ref = os.path.join(ROOT, "oracle", "_ref", "agrep")
try:
    import ctypes
    for readers in (1, 8, 16):
        os.environ["AGH_READERS"] = str(readers)
        q = A.Query(b"approximatematch", 2)
        fd = os.open(path, os.O_RDONLY)
        best = None
        for rep in range(3):
            os.lseek(fd, 0, os.SEEK_SET)
            t0 = time.time()
            r, _ = q.scan_fd(fd, flags=A.COUNT)
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
        os.close(fd)
        q.close()
        print("agh_scan_fd in-process, %d reader thread(s): %.3f s  %.2f GB/s -> %d" % (readers, best, n / 1e9 / best, r.n_matched))
    del os.environ["AGH_READERS"]
    print("host cores:", os.cpu_count())
    for name, exe in (("agrep-hip", cli), ("reference", ref)):
        if not os.path.exists(exe):
            continue
        for args in (["-V0", "-2", "-c"], ["-V0", "-2"]):
            best = None
            for rep in range(3):
                t0 = time.time()
                out = subprocess.run([exe] + args + ["approximatematch", path], stdout=subprocess.PIPE).stdout
                dt = time.time() - t0
                best = dt if best is None else min(best, dt)
            tag = out.split()[0].decode() if "-c" in args else "%d lines" % out.count(b"\n")
            print("%-10s %-12s %.3f s  %.2f GB/s  -> %s" % (name, " ".join(args), best, n / 1e9 / best, tag))
finally:
    os.unlink(path)
