"""Fused against two-kernel count-only scans when candidates are dense: patterns cut out of the
corpus itself (its vocabulary is small, so their grams are everywhere).  usage: [GiB, default 4]"""
import os, sys, time
os.environ.setdefault("AGH_ENV_LIVE", "1")   # switches are flipped between scans of one query
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
os.environ["AGH_FUSED_MIN_MB"] = "0"      # compare the two forms at every size (the default picks by size)
import torch
import agrep_amd as A
import bench as B

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gib * (1 << 30)) // 4096 * 4096
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
torch.cuda.synchronize()
head = bytes(t[:1 << 16].cpu().numpy())
pats = []
for off in (1000, 5000, 20000, 40000):
    s = head[off:off + 200].replace(b"\n", b" ")
    i = s.find(b" ") + 1
    pats.append(s[i:i + 16])
pats.append(B.PATTERN)
for pat in pats:
    for k in (0, 1, 2):
        q = A.Query(pat, k)
        info = q.info()
        line = "%-18r k=%d q=%d h=%d" % (pat, k, info["filter_q"], info["filter_h"])
        for f in ("0", "1"):
            os.environ["AGH_FUSED"] = f
            for _ in range(2):
                r = q.scan_device(t.data_ptr(), n, flags=A.COUNT)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                r = q.scan_device(t.data_ptr(), n, flags=A.COUNT)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            line += "  | fused=%s %.3f ms %.0f GB/s matched %d cand/MiB %.0f reruns %d fusedseg %d" % (
                f, dt * 1e3, n / 1e9 / dt, r.n_matched, r.n_candidates / (n / 2**20), r.lean_reruns, r.fused_segments)
        print(line, flush=True)
        q.close()
os.environ.pop("AGH_FUSED", None)
