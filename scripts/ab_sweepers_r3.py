"""A/B of the fused kernel's shape: sweeping waves per workgroup (build variants, AGH_LIB_PATH) x persistent
workgroups per CU (AGH_FUSED_BLOCKS), k = 2 (H = 2 samples) and k = 0, 64 GiB and 8 GiB, one process per library.
usage: AGH_LIB_PATH=<lib> scripts/ab_sweepers_r3.py <sweepers per workgroup of that build> [GiB] [steps]"""
import os, sys, time
os.environ.setdefault("AGH_ENV_LIVE", "1")   # switches are flipped between scans of one query
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
os.environ["AGH_FUSED_MIN_MB"] = "0"
import torch
import agrep_amd as A
import bench as B

ns = int(sys.argv[1])
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 64.0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n_all = int(gib * (1 << 30)) // 4096 * 4096
t = torch.empty(n_all, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n_all // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
torch.cuda.synchronize()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
for sz in [s for s in (64, 8) if s <= gib]:
    n = sz << 30
    for k in (2, 0):
        for per_cu in (1, 2, 3, 4):
            if ns * per_cu < 6 or ns * per_cu > 24:
                continue
            os.environ["AGH_FUSED_BLOCKS"] = str(per_cu * n_cu)
            q = A.Query(B.PATTERN, k)
            for _ in range(3):
                r = q.scan_device(t.data_ptr(), n, flags=A.COUNT, time_sweep=False, time_scan=False)
            torch.cuda.synchronize()
            tot = 0.0
            for _ in range(steps):
                t0 = time.perf_counter()
                r = q.scan_device(t.data_ptr(), n, flags=A.COUNT, time_sweep=False, time_scan=False)
                tot += time.perf_counter() - t0
            q.close()
            print("k=%d %5.1f GiB  %d sweepers x %d workgroups per CU = %2d sweeping waves per CU: avg %.4f ms  %.0f GB/s  matched %d fused %d"
                  % (k, sz, ns, per_cu, ns * per_cu, tot / steps * 1e3, n / 1e9 / (tot / steps), r.n_matched, r.fused_segments), flush=True)
