import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch, agrep_amd as A, _oracle as O
GiB=1<<30
n=4*GiB
t=torch.empty(n,dtype=torch.uint8,device='cuda')
t0=time.time(); planted=A.corpus_fill_device(t.data_ptr(), n//4096, seed=12345, variants=O.VARIANTS_C2, plant_period=500); torch.cuda.synchronize(); print("gen s",time.time()-t0, planted, sum(planted))
for i in range(3): print("probe ms", A.probe_read_ms(t.data_ptr(), n), "GB/s", n/1e6/A.probe_read_ms(t.data_ptr(), n))
for k in (0,1,2,3):
    q=A.Query(O.PATTERN_C2,k)
    for flags in (0, A.FORCE_FULLSCAN):
        for it in range(3):
            t0=time.time(); r=q.scan_device(t.data_ptr(), n, flags=flags); dt=time.time()-t0
        print("k",k,"engine",r.engine,"matched",r.n_matched,"records",r.n_records,"cand",r.n_candidates,"dev_ms %.3f sweep_ms %.3f wall_ms %.3f"%(r.device_ms,r.sweep_ms,dt*1e3),"GB/s dev %.0f sweep %.0f"%(n/1e6/r.device_ms, n/1e6/r.sweep_ms))
    q.close()
