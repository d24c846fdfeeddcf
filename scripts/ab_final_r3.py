"""Final round-3 A/B of the headline count-only scan, ONE process on one corpus: every step from the
round-2 kernel to the shipped default as an explicit row (the defaults changed during the round, so
scripts/ab_round3.py's unlabelled 'fused' row is the final form), plus the number of persistent
workgroups (AGH_FUSED_BLOCKS; shipped: 2 per CU of 6 sweeping + 2 verifying waves for H = 2 samples,
1 per CU of 8 + 2 waves else).
usage: scripts/ab_final_r3.py [total GiB, default 64] [steps, default 10]"""
import os, sys, time
os.environ.setdefault("AGH_ENV_LIVE", "1")   # switches are flipped between scans of one query
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
os.environ["AGH_FUSED_MIN_MB"] = "0"
import torch
import agrep_amd as A
import bench as B

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 64.0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_all = int(gib * (1 << 30)) // 4096 * 4096
t = torch.empty(n_all, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n_all // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
torch.cuda.synchronize()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
KEYS = ("AGH_FUSED", "AGH_FUSED_TAIL_MB", "AGH_FUSED_TAIL_KB", "AGH_SHAPE_H2", "AGH_FUSED_RANGE_KB", "AGH_FUSED_BLOCKS")


def run(label, k, n, env):
    for kk in KEYS:
        os.environ.pop(kk, None)
    os.environ.update(env)
    q = A.Query(B.PATTERN, k)
    info = q.info()
    for _ in range(3):
        r = q.scan_device(t.data_ptr(), n, flags=A.COUNT, time_sweep=False, time_scan=False)
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(steps):
        t0 = time.perf_counter()
        r = q.scan_device(t.data_ptr(), n, flags=A.COUNT, time_sweep=False, time_scan=False)
        tot += time.perf_counter() - t0
    q.close()
    print("k=%d %5.1f GiB %-44s q=%d h=%-2d avg %.4f ms  %.0f GB/s  matched %d cand %d reruns %d fused %d"
          % (k, n / 2**30, label, info["filter_q"], info["filter_h"], tot / steps * 1e3, n / 1e9 / (tot / steps),
             r.n_matched, r.n_candidates, r.lean_reruns, r.fused_segments), flush=True)
    return r.n_matched


OLD = {"AGH_SHAPE_H2": "0", "AGH_FUSED_TAIL_MB": "0"}
if len(sys.argv) > 3 and sys.argv[3] == "k1":       # k = 1 (4-byte samples every 4 bytes): workgroups per CU only
    for sz in [s for s in (64, 8) if s <= gib]:
        for label, env in (("shipped default", {}), ("1 workgroup per CU", {"AGH_FUSED_BLOCKS": str(1 * n_cu)}),
                           ("2 workgroups per CU", {"AGH_FUSED_BLOCKS": str(2 * n_cu)}),
                           ("3 workgroups per CU", {"AGH_FUSED_BLOCKS": str(3 * n_cu)}), ("two kernels", {"AGH_FUSED": "0"}),
                           ("shipped default again", {})):
            run(label, 1, sz << 30, env)
    sys.exit(0)
for sz in [s for s in (64, 8, 4, 2, 1) if s <= gib]:
    n = sz << 30
    for k in (2, 0):
        rows = [("two kernels, round-2 sample shape", dict(OLD, AGH_FUSED="0")),
                ("fused, round-2 shape and tickets", OLD),
                ("fused + small tickets at the end", {"AGH_SHAPE_H2": "0"}),
                ("fused + H=2 samples = shipped default", {}),
                ("two kernels, shipped shape", {"AGH_FUSED": "0"}),
                ("shipped, without the small tickets", {"AGH_FUSED_TAIL_MB": "0"}),
                ("shipped, 1 workgroup per CU", {"AGH_FUSED_BLOCKS": str(1 * n_cu)}),
                ("shipped, 2 workgroups per CU", {"AGH_FUSED_BLOCKS": str(2 * n_cu)}),
                ("shipped, 3 workgroups per CU", {"AGH_FUSED_BLOCKS": str(3 * n_cu)}),
                ("shipped default again", {})]
        if sz < 8:          # where does the fused kernel start to pay?  (AGH_FUSED_MIN_MB)
            rows = [rows[4], rows[3]]
        want = None
        for label, env in rows:
            got = run(label, k, n, env)
            want = got if want is None else want
            if got != want:
                print("MISMATCH %s: %d != %d" % (label, got, want), flush=True)
