"""A/B of the lean pipeline's knobs inside one process (same box, same corpus):
AGH_PART_MB (sweep launches per segment) x AGH_OVERLAP (verifier on a second stream), k = 2 and 0,
on the bench corpus.  usage: scripts/ab_pipeline.py [total GiB, default 64] [steps, default 10]"""
import os, sys, time
os.environ.setdefault("AGH_ENV_LIVE", "1")   # switches are flipped between scans of one query
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import bench as B

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 64.0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(gib * (1 << 30)) // 4096 * 4096
t = torch.empty(n, dtype=torch.uint8, device='cuda')
planted = A.corpus_fill_device(t.data_ptr(), n // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
torch.cuda.synchronize()
print("corpus %.0f GiB, planted %s" % (gib, planted), flush=True)
grid = [(0, 0), (0, 1), (4096, 0), (4096, 1), (2048, 1), (1024, 0), (1024, 1), (512, 1), (256, 1)]
if os.environ.get("AGH_AB_GRID"):            # e.g. "0:0,32768:1,16384:1,8192:1"
    grid = [tuple(int(x) for x in g.split(":")) for g in os.environ["AGH_AB_GRID"].split(",")]
for k in (2, 0):
    q = A.Query(B.PATTERN, k)
    for part, ov in grid:
        os.environ["AGH_PART_MB"] = str(part)
        os.environ["AGH_OVERLAP"] = str(ov)
        for timed in (False, True):
            fl = A.COUNT | (A.TIME_SWEEP if timed else 0)
            for _ in range(2):
                r = q.scan_device(t.data_ptr(), n, flags=fl, time_sweep=False, time_scan=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sw = 0.0; ln = 0
            for _ in range(steps):
                r = q.scan_device(t.data_ptr(), n, flags=fl, time_sweep=False, time_scan=False)
                sw += r.sweep_ms; ln += r.sweep_launches
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            print("k=%d part=%4d MiB overlap=%d events=%d: %.3f ms/scan  %.0f GB/s  matched %d  segs %d reruns %d  sweep launches %d avg %.4f ms (%.0f GB/s)"
                  % (k, part, ov, timed, dt * 1e3, n / 1e9 / dt, r.n_matched, r.n_segments, r.lean_reruns,
                     ln // steps, sw / max(ln, 1), (n * steps / max(ln, 1)) / 1e6 / (sw / max(ln, 1)) if ln else 0), flush=True)
    q.close()
rp = min(n, 8 << 30)
print("read probe %.0f GB/s" % (rp / 1e6 / min(A.probe_read_ms(t.data_ptr(), rp) for _ in range(3))))
