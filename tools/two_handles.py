"""Experiment: do two handles (two independent stream sets) on one GPU overlap each other's
latency-bound kernels?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from infercam_onnx_amd import nn, synth
W,H,B=640,480,32
weights=synth.synthetic_weights(); pri=synth.gen_priors(W,H)
jpegs=synth.synth_jpeg_pool(0,128,W,H)
for nh in (1,2,3):
  for thr in (16,32):
    ms=[nn.UltrafaceModel(nn.UltrafaceVariant.W640H480,0.5,0.5,max_batch=B,weights=weights,priors=pri,max_src=(W,H),host_threads=thr,det_cap=256) for _ in range(nh)]
    bs=[[m._prep_batch(jpegs[i*B:(i+1)*B]) for i in range(4)] for m in ms]
    for m,b in zip(ms,bs):
        for _ in range(3): m.wait(m.submit_jpeg_batch(b[0]),collect=False)
    N=60
    t00=time.perf_counter()
    infl=[]
    for s in range(N):
        if len(infl)>=2*nh:
            mm,t=infl.pop(0); mm.wait(t,collect=False)
        m=ms[s%nh]; infl.append((m,m.submit_jpeg_batch(bs[s%nh][(s//nh)%4])))
    for mm,t in infl: mm.wait(t,collect=False)
    el=time.perf_counter()-t00
    print('handles',nh,'threads',thr,'fps %.0f'%(N*B/el), 'ms/batch %.3f'%(el/N*1e3))
    for m in ms: m.close()
