import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import numpy as np
from infercam_onnx_amd import nn, synth
W,H,B=640,480,32
weights=synth.synthetic_weights(); pri=synth.gen_priors(W,H)
jpegs=synth.synth_jpeg_pool(0,64,W,H)
for prof in (False, True):
  for thr in (8,16,32,64):
    m=nn.UltrafaceModel(nn.UltrafaceVariant.W640H480,0.5,0.5,max_batch=B,weights=weights,priors=pri,max_src=(W,H),host_threads=thr,profile=prof,det_cap=256)
    bs=[m._prep_batch(jpegs[i*B:(i+1)*B]) for i in range(2)]
    for _ in range(3): m.wait(m.submit_jpeg_batch(bs[0]),collect=False)
    ts=[];tw=[]
    t00=time.perf_counter()
    N=30
    infl=[]
    for s in range(N):
        if len(infl)>=2:
            t0=time.perf_counter(); m.wait(infl.pop(0),collect=False); tw.append(time.perf_counter()-t0)
        t0=time.perf_counter(); infl.append(m.submit_jpeg_batch(bs[s%2])); ts.append(time.perf_counter()-t0)
    for t in infl: m.wait(t,collect=False)
    el=time.perf_counter()-t00
    print('profile',prof,'threads',thr,'fps %.0f'%(N*B/el),'submit ms %.3f'%(np.mean(ts)*1e3),'wait ms %.3f'%(np.mean(tw)*1e3))
    m.close()
