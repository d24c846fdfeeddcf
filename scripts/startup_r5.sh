#!/bin/bash
# Round 5: where the wall time of one `agrep-hip` process goes (AGH_TIMELINE=1 stamps of the library), on a 1 MiB
# file (BASELINE config 1 as a user meets it) and on a 4 GiB page-cache file; then the wall times themselves.
cd $GRAFT_REPO_ROOT
python - <<'PY'
import random, os
rng = random.Random(1)
words = ["approximate", "match", "pattern", "record", "delimiter", "needle", "haystack", "lorem", "ipsum", "dolor"]
with open("/tmp/c1.txt", "w") as f:
    n = 0
    while n < (1 << 20):
        line = " ".join(rng.choice(words) for _ in range(rng.randint(5, 14))) + "\n"
        f.write(line); n += len(line)
PY
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, agrep_amd as A, _oracle as O
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
t.cpu().numpy().tofile("/dev/shm/agh_r5_4g.txt")
PY
echo "== timeline, 1 MiB, -c haystack (second run: page cache and driver warm)"
agrep_amd/agrep-hip -c haystack /tmp/c1.txt > /dev/null
AGH_TIMELINE=1 agrep_amd/agrep-hip -c haystack /tmp/c1.txt
echo "== timeline, 4 GiB, -V0 -2 -c approximatematch"
agrep_amd/agrep-hip -V0 -2 -c approximatematch /dev/shm/agh_r5_4g.txt > /dev/null
AGH_TIMELINE=1 agrep_amd/agrep-hip -V0 -2 -c approximatematch /dev/shm/agh_r5_4g.txt
echo "== wall times (best of 5)"
python - <<'PY'
import os, subprocess, time
def t(label, cmd, reps=5, env=None):
    best = 1e9
    for _ in range(reps):
        t0 = time.time(); r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **(env or {}))); best = min(best, time.time() - t0)
    print("%-52s %.3f s  -> %s" % (label, best, r.stdout.strip().split("\n")[0][:40]), flush=True)
t("agrep-hip -c haystack, 1 MiB", ["agrep_amd/agrep-hip", "-c", "haystack", "/tmp/c1.txt"])
t("agrep-hip -1 -c haystack, 1 MiB", ["agrep_amd/agrep-hip", "-1", "-c", "haystack", "/tmp/c1.txt"])
t("agrep-hip -V0 -2 -c approximatematch, 4 GiB", ["agrep_amd/agrep-hip", "-V0", "-2", "-c", "approximatematch", "/dev/shm/agh_r5_4g.txt"])
t("agrep-hip -V0 -2 approximatematch, 4 GiB (records)", ["agrep_amd/agrep-hip", "-V0", "-2", "approximatematch", "/dev/shm/agh_r5_4g.txt"])
t("agrep_gpu (reference front end) -V0 -2, 4 GiB", ["oracle/_ref/agrep_gpu", "-V0", "-2", "approximatematch", "/dev/shm/agh_r5_4g.txt"], 3)
t("agrep_gpu -V0 -2 -c, 4 GiB", ["oracle/_ref/agrep_gpu", "-V0", "-2", "-c", "approximatematch", "/dev/shm/agh_r5_4g.txt"], 3)
t("agrep_gpu -V0 -2, 4 GiB, AGH_CLI_TEARDOWN=1", ["oracle/_ref/agrep_gpu", "-V0", "-2", "approximatematch", "/dev/shm/agh_r5_4g.txt"], 3, {"AGH_CLI_TEARDOWN": "1"})
t("agrep-hip -V0 -2 -c, 4 GiB, AGH_CLI_TEARDOWN=1", ["agrep_amd/agrep-hip", "-V0", "-2", "-c", "approximatematch", "/dev/shm/agh_r5_4g.txt"], 3, {"AGH_CLI_TEARDOWN": "1"})
t("agrep_gpu -V0 -2 > /dev/null, 4 GiB", ["sh", "-c", "oracle/_ref/agrep_gpu -V0 -2 approximatematch /dev/shm/agh_r5_4g.txt > /dev/null"], 3)
t("agrep-hip -V0 -2 > /dev/null, 4 GiB", ["sh", "-c", "agrep_amd/agrep-hip -V0 -2 approximatematch /dev/shm/agh_r5_4g.txt > /dev/null"], 3)
t("reference -c haystack, 1 MiB", ["oracle/_ref/agrep", "-V0", "-c", "haystack", "/tmp/c1.txt"])
PY
ls -la agrep_amd/libagrep_hip*.so
rm -f /dev/shm/agh_r5_4g.txt
