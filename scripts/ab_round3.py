"""Round-3 A/B of the headline count-only scan in ONE process on one corpus (same box, same bytes):
two kernels vs fused, small tickets at the end of the text (AGH_FUSED_TAIL_MB / _KB), and the H = 2 /
q = 4 sample shape (AGH_SHAPE_H2, read when the query is built) against H = 4 / q = 3 at k = 2.
usage: scripts/ab_round3.py [total GiB, default 64] [steps, default 10]
AGH_LIB_PATH=<variant .so> runs the same against another build (make -C agrep_amd/csrc FT_BITS=14)."""
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
planted = A.corpus_fill_device(t.data_ptr(), n_all // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
torch.cuda.synchronize()
print("lib %s corpus %.0f GiB planted %s" % (os.path.basename(A._ffi.LIB_PATH), gib, planted), flush=True)
KEYS = ("AGH_FUSED", "AGH_FUSED_TAIL_MB", "AGH_FUSED_TAIL_KB", "AGH_SHAPE_H2", "AGH_FUSED_RANGE_KB")


def run(label, k, n, env):
    for kk in KEYS:
        os.environ.pop(kk, None)
    os.environ.update(env)
    q = A.Query(B.PATTERN, k)
    info = q.info()
    for _ in range(3):
        r = q.scan_device(t.data_ptr(), n, flags=A.COUNT, time_sweep=False, time_scan=False)
    torch.cuda.synchronize()
    best = 1e9
    tot = 0.0
    for _ in range(steps):
        t0 = time.perf_counter()
        r = q.scan_device(t.data_ptr(), n, flags=A.COUNT, time_sweep=False, time_scan=False)
        dt = time.perf_counter() - t0
        best = min(best, dt)
        tot += dt
    q.close()
    print("k=%d %5.1f GiB %-34s q=%d h=%-2d avg %.4f ms  best %.4f ms  %.0f GB/s  matched %d cand %d reruns %d fused %d"
          % (k, n / 2**30, label, info["filter_q"], info["filter_h"], tot / steps * 1e3, best * 1e3,
             n / 1e9 / (tot / steps), r.n_matched, r.n_candidates, r.lean_reruns, r.fused_segments), flush=True)
    return r.n_matched


sizes = [s for s in (64, 16, 8, 4) if s <= gib]
for sz in sizes:
    n = sz << 30
    for k in (2, 0):
        want = run("two kernels", k, n, {"AGH_FUSED": "0"})
        cfgs = [("fused", {}),
                ("fused tail 256M/64K", {"AGH_FUSED_TAIL_MB": "256", "AGH_FUSED_TAIL_KB": "64"}),
                ("fused tail 512M/64K", {"AGH_FUSED_TAIL_MB": "512", "AGH_FUSED_TAIL_KB": "64"}),
                ("fused tail 1024M/64K", {"AGH_FUSED_TAIL_MB": "1024", "AGH_FUSED_TAIL_KB": "64"}),
                ("fused tail 512M/128K", {"AGH_FUSED_TAIL_MB": "512", "AGH_FUSED_TAIL_KB": "128"}),
                ("fused tail 1024M/128K", {"AGH_FUSED_TAIL_MB": "1024", "AGH_FUSED_TAIL_KB": "128"}),
                ("fused tail 512M/32K", {"AGH_FUSED_TAIL_MB": "512", "AGH_FUSED_TAIL_KB": "32"})]
        if k == 2:
            cfgs += [("two kernels H2", {"AGH_FUSED": "0", "AGH_SHAPE_H2": "1"}),
                     ("fused H2", {"AGH_SHAPE_H2": "1"}),
                     ("fused H2 tail 512M/64K", {"AGH_SHAPE_H2": "1", "AGH_FUSED_TAIL_MB": "512", "AGH_FUSED_TAIL_KB": "64"})]
        for label, env in cfgs:
            got = run(label, k, n, env)
            if got != want:
                print("MISMATCH %s: %d != %d" % (label, got, want), flush=True)
