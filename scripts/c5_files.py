"""BASELINE config 5 as a FILE job: N files x S MiB, 1024 patterns (8..12 bytes), k = 1 (--approx-f:
the reference ignores -# with -f) or exact, -l (file names only), through the C CLI with
--gpus min(4, devices): the files are dealt to the devices, every device scans its files with the
early exit of -l, the hit vector is reduced with RCCL inside the C-ABI (agh_reduce_file_hits_all).
One file in four holds planted patterns.  The expected list comes from resident scans of the same
bytes (agh_scan_device) before they are written.  End to end (page cache -> PCIe -> HBM), never the
roofline figure.   usage: scripts/c5_files.py [files, default 32] [MiB per file, default 1024] [k, default 1]"""
import os, random, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = random.Random(1024)
pats = set()
while len(pats) < 1024:
    pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(8, 12))))
pats = sorted(pats)
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
free = shutil.disk_usage(d).free
while nfiles > 4 and nfiles * (mib << 20) > free * 0.8:
    nfiles //= 2
pf = os.path.join(d, "agh_c5_pats.txt")
open(pf, "wb").write(b"\n".join(pats) + b"\n")
n = mib << 20
t = torch.empty(n, dtype=torch.uint8, device="cuda")
q = A.Query.multi(pats, k=k)
files, want = [], []
t_gen = time.time()
for f in range(nfiles):
    planted = f % 4 == 1
    # the other files get a corpus over an alphabet the patterns do not use much: with k = 1, 8-byte
    # patterns also match by chance every few MiB of lower-case text
    A.corpus_fill_device(t.data_ptr(), n // 4096, first_page=f * (n // 4096), seed=5,
                         variants=tuple(pats[:7]) if planted else (), plant_period=50000,
                         upper_permille=0 if planted else 1000)
    p = os.path.join(d, "agh_c5_file%02d.txt" % f)
    t.cpu().numpy().tofile(p)
    files.append(p)
    if q.scan_device(t.data_ptr(), n, flags=A.COUNT).n_matched:
        want.append(p)
q.close()
del t
torch.cuda.empty_cache()
ndev = A.device_count()
gpus = min(4, ndev)
print("c5 files: %d x %d MiB written in %.1f s, %d expected hits, %d device(s) visible" % (nfiles, mib, time.time() - t_gen, len(want), ndev), flush=True)
cli = os.path.join(ROOT, "agrep_amd", "agrep-hip")
kopt = (["--approx-f", "-%d" % k] if k else [])
for label, extra in (("--gpus %d (RCCL hit-vector reduce)" % gpus, ["--gpus", str(gpus)]), ("one-GPU path", [])):
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([cli] + extra + kopt + ["-V0", "-l", "-f", pf] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.time() - t0
    got = r.stdout.decode().split()
    print("c5 files %s: %d files x %d MiB, 1024 patterns (8..12 B), k=%d, -l: %.3f s wall (%.1f files/s, %.1f GB/s of file bytes; "
          "process start-up included), listed %d files, list == expected: %s, stderr: %r"
          % (label, nfiles, mib, k, dt, nfiles / dt, nfiles * n / 1e9 / dt, len(got), got == want, r.stderr[:120]), flush=True)
for p in files + [pf]:
    os.unlink(p)
