"""BASELINE config 5 as a FILE job on one GPU: N files, 1024 patterns, -l (file names only) through
the C CLI with --gpus 1 (files dealt to the devices, hit vector reduced with RCCL inside the C-ABI).
One file in four holds planted patterns; patterns are 8..12 bytes so that the others hold none.
End-to-end (page cache -> PCIe -> HBM), never the roofline figure.
usage: scripts/c5_files.py [files, default 8] [MiB per file, default 512]"""
import os, random, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rng = random.Random(1024)
pats = set()
while len(pats) < 1024:
    pats.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(8, 12))))
pats = sorted(pats)
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
pf = os.path.join(d, "agh_c5_pats.txt")
open(pf, "wb").write(b"\n".join(pats) + b"\n")
n = mib << 20
t = torch.empty(n, dtype=torch.uint8, device="cuda")
files, want = [], []
for f in range(nfiles):
    planted = f % 4 == 1
    A.corpus_fill_device(t.data_ptr(), n // 4096, first_page=f * (n // 4096), seed=5,
                         variants=tuple(pats[:7]) if planted else (), plant_period=50000)
    p = os.path.join(d, "agh_c5_file%02d.txt" % f)
    t.cpu().numpy().tofile(p)
    files.append(p)
    if planted:
        want.append(p)
del t
cli = os.path.join(ROOT, "agrep_amd", "agrep-hip")
for label, extra in (("--gpus 1 (RCCL hit-vector reduce)", ["--gpus", "1"]), ("one-GPU path", [])):
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([cli] + extra + ["-V0", "-l", "-f", pf] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.time() - t0
    got = r.stdout.decode().split()
    print("c5 files %s: %d files x %d MiB, 1024 patterns (8..12 B), -l: %.3f s wall (%.1f files/s, %.1f GB/s of file bytes; "
          "process start-up included), listed %d files, expected list: %s, stderr: %r"
          % (label, nfiles, mib, dt, nfiles / dt, nfiles * n / 1e9 / dt, len(got), got == want, r.stderr[:120]), flush=True)
for p in files + [pf]:
    os.unlink(p)
