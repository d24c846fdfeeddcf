"""Table engine ('#', ';', ',' patterns; agh_table.hip) on the 4 GiB C2 corpus: two streams per lane
(k_tablescan_fast2, tables of M <= 15 positions) against one (AGH_TF_PACK2=0) and against the exact one-kernel
form (AGH_FS_FAST=0).  Patterns go through the library's own compiler (agh_query_pattern).
usage: scripts/perf_table_r4.py [GiB, default 4]"""
import os, sys
os.environ.setdefault("AGH_ENV_LIVE", "1")
os.environ.setdefault("AGH_TF_FAST_MIN_MB", "0")   # switches are flipped between scans of one query
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


cases = [(b"approx#match", 0), (b"approx#match", 1), (b"approx#match", 2), (b"appr#mate#ch", 1), (b"scar,cat", 0), (b"match;approx", 0),
         (b"match;approx", 1), (b"approxi;matematch", 0), (b"approxi;matematch", 1)]
quick = os.environ.get("PERF_TABLE_QUICK") == "1"        # (under rocprofv3: two patterns, the default form only)
if quick:
    cases = [(b"approx#match", 0), (b"approx#match", 1)]
forms = (("two streams", {}), ("one stream", {"AGH_TF_PACK2": "0"}), ("k_tablescan", {"AGH_FS_FAST": "0"}))
for pat, k in cases:
    row = []
    for name, env in (forms[:2] if quick else forms):
        for key in ("AGH_TF_PACK2", "AGH_FS_FAST"):
            os.environ.pop(key, None)
        os.environ.update(env)
        with A.Query.pattern(pat, k) as q:
            ms_n, r_n = med(q, 0, 3)
            ms_c, r_c = med(q, A.COUNT, 5)
        row.append("%s: numbered %.3f ms %.0f GB/s, count-only %.3f ms %.0f GB/s (matched %d/%d)"
                   % (name, ms_n, n / 1e6 / ms_n, ms_c, n / 1e6 / ms_c, r_n.n_matched, r_c.n_matched))
    print("table '%s' k=%d  %s" % (pat.decode(), k, " | ".join(row)), flush=True)
for key in ("AGH_TF_PACK2", "AGH_FS_FAST"):
    os.environ.pop(key, None)
