"""Does the sweep kernel time depend on what ran just before it (clock / power state)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch
import agrep_amd as A
import _oracle as O
n = 4 << 30
t = torch.empty(n, dtype=torch.uint8, device='cuda')
A.corpus_fill_device(t.data_ptr(), n // 4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500)
q = A.Query(O.PATTERN_C2, 2)
def run(label, pause, reps=8):
    xs = []
    for i in range(reps):
        if pause: time.sleep(pause)
        r = q.scan_device(t.data_ptr(), n)
        xs.append(r.sweep_ms)
    print(label, " ".join("%.3f" % x for x in xs))
run("back-to-back   ", 0)
run("sleep 2 ms     ", 0.002)
run("sleep 50 ms    ", 0.05)
run("sleep 500 ms   ", 0.5, 4)
run("back-to-back   ", 0)
for fl, lab in ((A.FORCE_FULLSCAN, "fullscan-sweep0"),):
    xs = []
    for i in range(5):
        r = q.scan_device(t.data_ptr(), n, flags=fl); xs.append(r.sweep_ms)
    print(lab, " ".join("%.3f" % x for x in xs))
print("probe", [round(A.probe_read_ms(t.data_ptr(), n), 3) for _ in range(5)])
