"""Round 5: the record-returning paths under rocprofv3 --kernel-trace --stats.
   c2: m=16 k=2, 4 GiB, agh_scan_device_emit (numbered scan + ordered list + bounds + gather + one copy)
   c3: m=48 k=3 -i, 16 GiB, numbered scan (two segments)
usage: scripts/prof_records_r5.py [c2] [c3] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import bench as B

which = [a for a in sys.argv[1:] if not a.isdigit()] or ["c2", "c3"]
reps = int(next((a for a in sys.argv[1:] if a.isdigit()), "6"))
buf = torch.empty(16 << 30, dtype=torch.uint8, device='cuda')
F = A.TIME_SWEEP | A.TIME_SCAN
if "c2" in which:
    n = 4 << 30
    A.corpus_fill_device(buf.data_ptr(), n // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
    with A.Query(B.PATTERN, 2) as q:
        xs, ds = [], []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res, batches = q.scan_device_emit(buf.data_ptr(), n, flags=F, summarize=True)
            xs.append(time.perf_counter() - t0); ds.append(res.device_ms)
        print("c2 emit 4 GiB: wall %.3f ms (min %.3f) device scan %.3f ms sweep %.3f ms records %d bytes %d calls %d"
              % (sorted(xs)[len(xs) // 2] * 1e3, min(xs) * 1e3, sorted(ds)[len(ds) // 2], res.sweep_ms, sum(b[0] for b in batches),
                 sum(b[1] for b in batches), len(batches)))
        xs = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = q.scan_device(buf.data_ptr(), n, flags=F)
            xs.append(time.perf_counter() - t0)
        print("c2 numbered count 4 GiB: wall %.3f ms device %.3f sweep %.3f matched %d" % (sorted(xs)[len(xs) // 2] * 1e3, res.device_ms, res.sweep_ms, res.n_matched))
if "c3" in which:
    n = 16 << 30
    pat, vs = B.c3_pattern_and_variants()
    A.corpus_fill_device(buf.data_ptr(), n // 4096, seed=9, variants=vs, plant_period=500, upper_permille=500)
    with A.Query(pat, 3, nocase=True) as q:
        for fl, lab in ((A.COUNT | F, "count-only"), (F, "numbered")):
            xs = []
            for _ in range(reps):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                res = q.scan_device(buf.data_ptr(), n, flags=fl)
                xs.append(time.perf_counter() - t0)
            w = sorted(xs)[len(xs) // 2]
            print("c3 %s 16 GiB: wall %.3f ms (%.0f GB/s) device %.3f sweep %.3f (%d launches) matched %d segments %d"
                  % (lab, w * 1e3, n / 1e9 / w, res.device_ms, res.sweep_ms, res.sweep_launches, res.n_matched, res.n_segments))
