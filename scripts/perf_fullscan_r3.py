"""Full-scan engine, fast form (k_fullscan_fast + k_fullscan_replay) against the exact one-kernel form
(AGH_FS_FAST=0) on the 4 GiB C2 corpus: the queries that really land there (short cores) and the
headline pattern forced onto it.  usage: scripts/perf_fullscan_r3.py [GiB, default 4]"""
import os, sys
os.environ.setdefault("AGH_ENV_LIVE", "1")   # switches are flipped between scans of one query
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


cases = [(B.PATTERN, 0, False), (B.PATTERN, 1, False), (B.PATTERN, 2, False), (B.PATTERN, 3, False), (B.PATTERN, 4, False),
         (b"approxim", 2, False), (b"match", 1, False), (b"approximate", 3, False),
         (b"approximatematchapproxim", 2, False), (b"approximatematchapproximatematchapproximatemat", 3, True)]
for pat, k, nocase in cases:
    row = []
    for fast in ("1", "0"):
        os.environ["AGH_FS_FAST"] = fast
        with A.Query(pat, k, nocase=nocase) as q:
            ms_n, r_n = med(q, A.FORCE_FULLSCAN)
            ms_c, r_c = med(q, A.FORCE_FULLSCAN | A.COUNT)
        row.append((ms_n, ms_c, r_n.n_matched, r_c.n_matched, r_c.n_candidates))
    print("m=%2d k=%d%s  fast: numbered %.3f ms %.0f GB/s, count-only %.3f ms %.0f GB/s (matched %d/%d, replayed pieces %d) | "
          "exact: numbered %.3f ms %.0f GB/s, count-only %.3f ms %.0f GB/s (matched %d/%d)"
          % (len(pat), k, " -i" if nocase else "", row[0][0], n / 1e6 / row[0][0], row[0][1], n / 1e6 / row[0][1], row[0][2], row[0][3],
             row[0][4], row[1][0], n / 1e6 / row[1][0], row[1][1], n / 1e6 / row[1][1], row[1][2], row[1][3]), flush=True)

# table engine ('#', ';', ',' on the reference's own tables): fast form (k_tablescan_fast + k_table_replay) vs k_tablescan
import json
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "pattern_language.json")))["cases"]
for c in gold:
    if c["pattern"] in ("approx#match", "approxi;matematch", "scar,cat") and c["k"] <= 1:
        tb = c["tables"]
        M = tb["D_endpos"].bit_length()
        row = []
        for fast in ("1", "0"):
            os.environ["AGH_FS_FAST"] = fast
            q = A.Query.from_maskgen(tb["Mask"], tb["Init0"], tb["Init1"], tb["NO_ERR_MASK"], tb["endposition"],
                                     tb["D_endpos"], M, b"\n", c["k"], tb["AND"])
            ms_n, r_n = med(q, 0, 3)
            ms_c, r_c = med(q, A.COUNT, 3)
            q.close()
            row.append((ms_n, ms_c, r_n.n_matched, r_c.n_matched))
        print("table '%s' k=%d  fast: numbered %.3f ms %.0f GB/s, count-only %.3f ms %.0f GB/s (matched %d/%d) | "
              "k_tablescan: numbered %.3f ms %.0f GB/s, count-only %.3f ms %.0f GB/s (matched %d/%d)"
              % (c["pattern"], c["k"], row[0][0], n / 1e6 / row[0][0], row[0][1], n / 1e6 / row[0][1], row[0][2], row[0][3],
                 row[1][0], n / 1e6 / row[1][0], row[1][1], n / 1e6 / row[1][1], row[1][2], row[1][3]), flush=True)
os.environ.pop("AGH_FS_FAST", None)
