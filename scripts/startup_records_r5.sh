#!/bin/bash
# Round 5: timeline of one agrep-hip process that PRINTS the matched records of a 4 GiB page-cache file.
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
for i in 1 2; do agrep_amd/agrep-hip -V0 -2 approximatematch /dev/shm/agh_r5_4g.txt > /dev/null; done
echo "== records > /dev/null"
AGH_TIMELINE=1 agrep_amd/agrep-hip -V0 -2 approximatematch /dev/shm/agh_r5_4g.txt 2>&1 > /dev/null | grep -v "lean_run\|worker:" | head -40
echo "== count"
AGH_TIMELINE=1 agrep_amd/agrep-hip -V0 -2 -c approximatematch /dev/shm/agh_r5_4g.txt 2>&1 > /dev/null | grep -v "lean_run\|worker:" | head -20
python - <<'PY'
import os, subprocess, time
def t(label, cmd, reps=7):
    xs = []
    for _ in range(reps):
        t0 = time.time(); subprocess.run(cmd, shell=True); xs.append(time.time() - t0)
    xs.sort(); print("%-50s best %.3f median %.3f" % (label, xs[0], xs[len(xs)//2]), flush=True)
t("-c", "agrep_amd/agrep-hip -V0 -2 -c approximatematch /dev/shm/agh_r5_4g.txt > /dev/null")
t("records", "agrep_amd/agrep-hip -V0 -2 approximatematch /dev/shm/agh_r5_4g.txt > /dev/null")
t("-c", "agrep_amd/agrep-hip -V0 -2 -c approximatematch /dev/shm/agh_r5_4g.txt > /dev/null")
t("records", "agrep_amd/agrep-hip -V0 -2 approximatematch /dev/shm/agh_r5_4g.txt > /dev/null")
PY
