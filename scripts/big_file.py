"""File larger than one device segment (8 GiB): agh_scan_fd -> two segments, parallel readers;
count must equal the resident scan and the planted number."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O
n = int(8.5 * (1 << 30)) // 4096 * 4096
t = torch.empty(n, dtype=torch.uint8, device='cuda')
planted = A.corpus_fill_device(t.data_ptr(), n // 4096, seed=77, variants=O.VARIANTS_C2, plant_period=500)
q = A.Query(O.PATTERN_C2, 2)
r_dev = q.scan_device(t.data_ptr(), n, flags=A.COUNT)
path = "/dev/shm/agh_big.txt"
try:
    with open(path, "wb") as f:
        step = 1 << 30
        for off in range(0, n, step):
            f.write(t[off:off + step].cpu().numpy().tobytes())
    del t
    torch.cuda.empty_cache()
    fd = os.open(path, os.O_RDONLY)
    t0 = time.time()
    r_fd, _ = q.scan_fd(fd, flags=A.COUNT)
    dt = time.time() - t0
    os.lseek(fd, 0, os.SEEK_SET)
    r_fd2, ms = q.scan_fd(fd, cap=400000)
    os.close(fd)
    print("8.5 GiB file: resident count %d, file count %d (%.2f s, %.1f GB/s), with records %d stored %d, planted %d"
          % (r_dev.n_matched, r_fd.n_matched, dt, n / 1e9 / dt, r_fd2.n_matched, len(ms), sum(planted)))
    assert r_dev.n_matched == r_fd.n_matched == r_fd2.n_matched == len(ms) == sum(planted)
    assert all(ms[i][0] < ms[i + 1][0] and ms[i][2] < ms[i + 1][2] for i in range(len(ms) - 1))
    # spot-check records of both segments against the file
    with open(path, "rb") as f:
        for s_, e_, idx in ms[:50] + ms[-50:]:
            f.seek(s_)
            rec = f.read(e_ - s_)
            assert b"\n" not in rec and O.dp_best(O.PATTERN_C2, rec) <= 2, (s_, e_, rec)
            if s_:
                f.seek(s_ - 1)
                assert f.read(1) == b"\n"
    print("ok")
finally:
    if os.path.exists(path):
        os.unlink(path)
