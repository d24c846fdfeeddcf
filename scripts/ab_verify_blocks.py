"""A/B of the verifier's grid (AGH_VERIFY_BLOCKS; 0 = one workgroup per group of 8 slices) on the
bench corpus, inside one process.  usage: scripts/ab_verify_blocks.py [GiB, default 64]"""
import os, sys, time
os.environ.setdefault("AGH_ENV_LIVE", "1")   # switches are flipped between scans of one query
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import bench as B
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 64.0
n = int(gib * (1 << 30)) // 4096 * 4096
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
torch.cuda.synchronize()
for k in (2, 0):
    q = A.Query(B.PATTERN, k)
    for rnd in range(2):
        for blocks in (0, 1024, 2048, 4096, 8192, 16384):
            os.environ["AGH_VERIFY_BLOCKS"] = str(blocks)
            for _ in range(2):
                r = q.scan_device(t.data_ptr(), n, flags=A.COUNT, time_sweep=False, time_scan=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                r = q.scan_device(t.data_ptr(), n, flags=A.COUNT, time_sweep=False, time_scan=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 8
            print("k=%d verify blocks cap %5d: %.3f ms/scan  %.0f GB/s  matched %d" % (k, blocks, dt * 1e3, n / 1e9 / dt, r.n_matched), flush=True)
    q.close()
