"""Table engine under delimiters of several bytes / a folded letter on 4 GiB: the fast form (branch-free kernel with the
record ends from the delimiter-end bitmap + exact replay, round 5) against the exact one-kernel form (AGH_FS_FAST=0).
usage: scripts/perf_table_mb_r5.py [GiB]"""
import os, sys
os.environ.setdefault("AGH_ENV_LIVE", "1")
os.environ.setdefault("AGH_TF_FAST_MIN_MB", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import bench as B
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gib * (1 << 30))
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)


def med(q, flags, reps=5):
    xs = []
    for _ in range(reps):
        r = q.scan_device(t.data_ptr(), n, flags=flags)
        xs.append(r.device_ms)
    return sorted(xs)[reps // 2], r


# (delimiters that occur in the synthetic corpus: "e " every ~130 bytes, "s\n" every ~1.8 KB, the letter z / Z every ~50)
for pat, k, delim, nocase in ((b"approx#match", 0, b"\n", False), (b"approx#match", 0, b"e ", False), (b"approx#match", 1, b"e ", False),
                              (b"approx#match", 2, b"e ", False), (b"approx#match", 1, b"s\n", False), (b"approx#match", 1, b"z", True),
                              (b"match,approx", 1, b"e ", False), (b"approximate#match", 2, b"e ", False)):
    row = []
    for name, env in (("fast form", {}), ("k_tablescan", {"AGH_FS_FAST": "0"})):
        os.environ.pop("AGH_FS_FAST", None)
        os.environ.update(env)
        with A.Query.pattern(pat, k, nocase=nocase, delim=delim) as q:
            ms_n, r_n = med(q, A.TIME_SCAN, 3)
            ms_c, r_c = med(q, A.COUNT | A.TIME_SCAN, 5)
        row.append("%s: numbered %.3f ms %.0f GB/s, count-only %.3f ms %.0f GB/s (matched %d/%d)"
                   % (name, ms_n, n / 1e6 / ms_n, ms_c, n / 1e6 / ms_c, r_n.n_matched, r_c.n_matched))
    print("table '%s' k=%d -d %r%s  %s" % (pat.decode(), k, delim, " -i" if nocase else "", " | ".join(row)), flush=True)
os.environ.pop("AGH_FS_FAST", None)
