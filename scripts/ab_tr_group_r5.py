"""k_table_replay: tiles per wave (AGH_TR_GROUP = 1, 2, 4, 8, 16), whole count-only scans of 4 GiB."""
import os, sys
os.environ.setdefault("AGH_ENV_LIVE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import bench as B
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
for pat, k, delim in ((b"approx#match", 1, b"\n"), (b"approx#match", 1, b"e "), (b"approx#match", 1, b"s\n"), (b"approx#match", 2, b"\n")):
    row = []
    for g in ("1", "2", "4", "8", "16"):
        os.environ["AGH_TR_GROUP"] = g
        with A.Query.pattern(pat, k, delim=delim) as q:
            xs = sorted(q.scan_device(t.data_ptr(), n, flags=A.COUNT | A.TIME_SCAN).device_ms for _ in range(7))
            xn = sorted(q.scan_device(t.data_ptr(), n, flags=A.TIME_SCAN).device_ms for _ in range(5))
        row.append("%s: %.3f / %.3f" % (g, xs[3], xn[2]))
    print("'%s' k=%d -d %r  count-only / numbered ms by group  %s" % (pat.decode(), k, delim, "  ".join(row)), flush=True)
