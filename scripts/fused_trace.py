"""(needs the diagnostics build: make -C agrep_amd/csrc EXP=1.)  When did each wave of the fused kernel stop
sweeping and when did it leave?  usage: scripts/fused_trace.py [GiB, default 8]"""
import ctypes, os, sys
os.environ.setdefault("AGH_ENV_LIVE", "1")   # switches are flipped between scans of one query
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
os.environ["AGH_FUSED_MIN_MB"] = "0"
import numpy as np, torch
import agrep_amd as A
from agrep_amd import _ffi
import bench as B
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(gib * (1 << 30)) // 4096 * 4096
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
q = A.Query(B.PATTERN, 2)
for _ in range(3):
    r = q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SWEEP)
buf = (ctypes.c_uint64 * (4 * 8192))()
lib = _ffi.lib()
assert lib.agh_debug_fused_trace(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4, 8192).astype(np.int64)
nblk = min(1024, (n // (256 << 10) + 3) // 4)
idx = np.array([b * 8 + w for b in range(nblk) for w in range(6)])
sw = np.array([b * 8 + w for b in range(nblk) for w in range(4)])
t0 = a[0][idx].min()
us = lambda x: (x - t0) / 100.0
print("kernel %.1f us by events; waves start %.1f..%.1f us" % (r.sweep_ms * 1e3, us(a[0][idx]).min(), us(a[0][idx]).max()))
e1 = us(a[1][sw]); e2 = us(a[2][idx])
for nm, e in (("sweeping waves stop streaming", e1), ("waves leave the kernel", e2)):
    print("%-32s min %.1f  p10 %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f us" % (
        nm, e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max()))

cnt = a[3][sw]; dur = (a[1][sw] - a[0][sw]) / 100.0
per = dur / np.maximum(cnt, 1)
print("ranges per sweeping wave: min %d median %d max %d; us per range: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (
    cnt.min(), np.median(cnt), cnt.max(), per.min(), np.percentile(per, 10), np.median(per), np.percentile(per, 90), per.max()))
pb = per.reshape(-1, 4)
print("spread of us/range inside a workgroup (max-min over its 4 waves): median %.1f p90 %.1f; spread of workgroup means: p10 %.1f median %.1f p90 %.1f" % (
    np.median(pb.max(1) - pb.min(1)), np.percentile(pb.max(1) - pb.min(1), 90), np.percentile(pb.mean(1), 10), np.median(pb.mean(1)), np.percentile(pb.mean(1), 90)))
started = us(a[0][idx].reshape(-1, 6)[:, 0])
print("workgroups that started later than 50 us: %d of %d" % ((started > 50).sum(), nblk))
xcd = np.arange(nblk) % 8
m = pb.mean(1)
print("mean us/range by workgroup index mod 8 (XCD):", " ".join("%.1f" % m[xcd == i].mean() for i in range(8)))
