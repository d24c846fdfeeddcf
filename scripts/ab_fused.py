"""A/B of the count-only pipeline inside one process (same box, same corpus): AGH_FUSED=0 (k_sweep, then
k_verify) against AGH_FUSED=1 (sweep + verify in one kernel, agh_fused.hip), k = 2 and 0 on the bench
corpus, plus ragged sizes for the tail path.  usage: scripts/ab_fused.py [total GiB, default 64] [steps]
AGH_LIB_PATH=<variant .so> runs the same against another build (e.g. make -C agrep_amd/csrc FT_BITS=14)."""
import os, sys, time
os.environ.setdefault("AGH_ENV_LIVE", "1")   # switches are flipped between scans of one query
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
os.environ["AGH_FUSED_MIN_MB"] = "0"      # compare the two forms at every size (the default picks by size)
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
modes = os.environ.get("AGH_AB_MODES", "0,1,0,1").split(",")   # fused flag [+ KiB per ticket]
for k in (2, 0):
    q = A.Query(B.PATTERN, k)
    # ragged sizes first: both pipelines must agree byte for byte on the count
    for nn in (n, (1 << 30) + 12345, (1 << 20) + 1023, 300000, 5000, 1024, 777):
        if nn > n: continue
        got = []
        for f in ("0", "1"):
            os.environ["AGH_FUSED"] = f
            r = q.scan_device(t.data_ptr(), nn, flags=A.COUNT, time_sweep=False, time_scan=False)
            got.append((r.n_matched, r.n_candidates, r.lean_reruns))
        print("k=%d n=%d  two-kernel %s  fused %s  %s" % (k, nn, got[0], got[1], "OK" if got[0][0] == got[1][0] else "MISMATCH"), flush=True)
    for f in modes:
        os.environ["AGH_FUSED"] = f[0]
        os.environ.pop("AGH_FUSED_RANGE_KB", None)
        if len(f) > 1: os.environ["AGH_FUSED_RANGE_KB"] = f[1:]          # KiB per ticket
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
            print("k=%d fused=%s events=%d: %.3f ms/scan  %.0f GB/s  matched %d cand %d reruns %d  kernel avg %.4f ms"
                  % (k, f, timed, dt * 1e3, n / 1e9 / dt, r.n_matched, r.n_candidates, r.lean_reruns, sw / max(ln, 1)), flush=True)
    q.close()
