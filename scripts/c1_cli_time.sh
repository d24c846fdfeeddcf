#!/bin/bash
# BASELINE config 1 as a user meets it: the command-line tools on a 1 MiB file, wall time per process.
cd $GRAFT_REPO_ROOT
python - <<'PY'
import random, subprocess, time, os, ctypes
rng = random.Random(1)
words = ["approximate", "match", "pattern", "record", "delimiter", "needle", "haystack", "lorem", "ipsum", "dolor"]
with open("/tmp/c1.txt", "w") as f:
    n = 0
    while n < (1 << 20):
        line = " ".join(rng.choice(words) for _ in range(rng.randint(5, 14))) + "\n"
        f.write(line); n += len(line)
def t(label, cmd, env=None, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
        best = min(best, time.time() - t0)
    print("%-44s %.3f s  -> %s" % (label, best, r.stdout.strip().split("\n")[0][:40]), flush=True)
t("agrep-hip -c haystack (k=0)", ["agrep_amd/agrep-hip", "-c", "haystack", "/tmp/c1.txt"])
t("agrep-hip -1 -c haystack", ["agrep_amd/agrep-hip", "-1", "-c", "haystack", "/tmp/c1.txt"])
t("agrep-hip --gpus 1 -c haystack (host sum)", ["agrep_amd/agrep-hip", "--gpus", "1", "-c", "haystack", "/tmp/c1.txt"])
t("agrep-hip --gpus 1 -l haystack (host sum)", ["agrep_amd/agrep-hip", "--gpus", "1", "-l", "haystack", "/tmp/c1.txt"])
t("agrep-hip --gpus 1 -c, AGH_CLI_RCCL=1", ["agrep_amd/agrep-hip", "--gpus", "1", "-c", "haystack", "/tmp/c1.txt"], {"AGH_CLI_RCCL": "1"})
t("agrep-hip -c 'h[a-c]ystack' (compiled class)", ["agrep_amd/agrep-hip", "-c", "h[a-c]ystack", "/tmp/c1.txt"])
t("reference -c haystack (k=0)", ["oracle/_ref/agrep", "-V0", "-c", "haystack", "/tmp/c1.txt"])
t("reference -1 -c haystack", ["oracle/_ref/agrep", "-V0", "-1", "-c", "haystack", "/tmp/c1.txt"])
t("agrep-hip, HIP_ENABLE_DEFERRED_LOADING=0", ["agrep_amd/agrep-hip", "-c", "haystack", "/tmp/c1.txt"], {"HIP_ENABLE_DEFERRED_LOADING": "0"})
t("agrep-hip --version-like no-op (-V)", ["agrep_amd/agrep-hip", "-V"])
t0 = time.time(); h = ctypes.CDLL("libamdhip64.so"); t1 = time.time()
n = ctypes.c_int(); h.hipGetDeviceCount(ctypes.byref(n)); t2 = time.time()
p = ctypes.c_void_p(); h.hipMalloc(ctypes.byref(p), 1 << 20); t3 = time.time()
print("in one process: dlopen libamdhip64 %.3f s, hipGetDeviceCount %.3f s, first hipMalloc %.3f s" % (t1 - t0, t2 - t1, t3 - t2))
t4 = time.time(); a = ctypes.CDLL("agrep_amd/libagrep_hip.so"); t5 = time.time()
print("dlopen libagrep_hip.so (runtime up) %.3f s" % (t5 - t4))
PY
