#!/bin/bash
# Round 5: one agrep-hip process on a 4 GiB page-cache file -- pinned chunk size of the staging ring (AGH_STAGE_CHUNK_MB)
# and the timeline of the shipped setting; all on ONE box (boxes differ by tens of milliseconds).
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, agrep_amd as A, _oracle as O
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
t.cpu().numpy().tofile("/dev/shm/agh_r5_4g.txt")
PY
python - <<'PY'
import os, subprocess, time
def t(label, cmd, reps=7, env=None):
    xs = []
    for _ in range(reps):
        t0 = time.time(); r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **(env or {}))); xs.append(time.time() - t0)
    xs.sort()
    print("%-60s best %.3f  median %.3f s  -> %s" % (label, xs[0], xs[len(xs) // 2], r.stdout.strip().split("\n")[0][:24]), flush=True)
cmd = ["agrep_amd/agrep-hip", "-V0", "-2", "-c", "approximatematch", "/dev/shm/agh_r5_4g.txt"]
for rnd in range(2):
    for mb in ("32", "16", "8", "64"):
        t("agrep-hip -c 4 GiB, AGH_STAGE_CHUNK_MB=%s (round %d)" % (mb, rnd), cmd, 7, {"AGH_STAGE_CHUNK_MB": mb})
PY
AGH_TIMELINE=1 agrep_amd/agrep-hip -V0 -2 -c approximatematch /dev/shm/agh_r5_4g.txt 2>&1 | head -16
