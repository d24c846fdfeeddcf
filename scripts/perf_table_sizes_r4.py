"""Table engine by input size: where two streams per lane (512 KiB of text per wave) stop paying against one stream
(256 KiB per wave) and against the exact kernel (64 KiB per wave) -- prefixes of the 4 GiB C2 corpus, count-only,
'approx#match'.  usage: scripts/perf_table_sizes_r4.py"""
import os, sys
os.environ.setdefault("AGH_ENV_LIVE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import bench as B

N = 4 << 30
t = torch.empty(N, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), N // 4096, seed=B.SEED, variants=B.VARIANTS, plant_period=500)
forms = (("two 4K", {"AGH_TF_FAST_MIN_MB": "0", "AGH_TF_CHUNK": "4096"}), ("two by size", {"AGH_TF_FAST_MIN_MB": "0", "AGH_TF_CHUNK": "0"}),
         ("one by size", {"AGH_TF_FAST_MIN_MB": "0", "AGH_TF_PACK2": "0", "AGH_TF_CHUNK": "0"}),
         ("exact", {"AGH_FS_FAST": "0"}), ("default", {}))
for k in (0, 1, 2):
    with A.Query.pattern(b"approx#match", k) as q:
        for mib in (4, 16, 64, 128, 256, 512, 1024, 4096):
            n = mib << 20
            row = []
            for name, env in forms:
                for key in ("AGH_TF_PACK2", "AGH_FS_FAST", "AGH_TF_FAST_MIN_MB", "AGH_TF_CHUNK"):
                    os.environ.pop(key, None)
                os.environ.update(env)
                xs = []
                for _ in range(5):
                    r = q.scan_device(t.data_ptr(), n, flags=A.COUNT)
                    xs.append(r.device_ms)
                ms = sorted(xs)[2]
                row.append("%s %.3f ms %5.0f GB/s" % (name, ms, n / 1e6 / ms))
            print("k=%d %5d MiB  %s  (matched %d)" % (k, mib, " | ".join(row), r.n_matched), flush=True)
