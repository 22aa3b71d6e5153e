#!/usr/bin/env python3
"""Round 6: where should batch n + 1's network wait for batch n (csrc/pipeline_gate.cpp)?  One process, one frame pool; per
setting of UFD_GATE_LAYER (measurement build: make EXPERIMENTS=1, UFD_LIBRARY=.../libufacehip_exp.so) a fresh handle, then
alternating over the settings: the driver's sample (5 warm-up steps, drain, 20 timed steps, drain -- bench.py's timed region)
several times and a 300-step steady-state run.  Prints frames/s per setting: median of the 20-step samples | steady state.
Usage on the box:  UFD_LIBRARY=$PWD/infercam_onnx_amd/libufacehip_exp.so python3 tools/ab/r6_gate.py [layers ...]   (-1 = gate off)"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from infercam_onnx_amd import nn, synth  # noqa: E402

B, DEPTH = 32, 6
layers = [int(a) for a in sys.argv[1:]] or [-1, 0, 4, 8, 12, 24, 34]
weights, priors = synth.synthetic_weights(), synth.gen_priors(640, 480)
jpegs = synth.synth_jpeg_pool(0, 256, 640, 480, quality=90, subsampling="4:2:0")


def run_steps(m, bts, k):
    infl = []
    for s in range(k):
        if len(infl) >= DEPTH:
            m.wait(infl.pop(0), collect=False)
        infl.append(m.submit_jpeg_batch(bts[s % len(bts)]))
    for t in infl:
        m.wait(t, collect=False)


models = {}
for L in layers:
    os.environ["UFD_GATE_LAYER"] = str(L)
    m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, max_batch=B, weights=weights, priors=priors, max_src=(640, 480), det_cap=256)
    bts = [m._prep_batch(jpegs[i * B:(i + 1) * B]) for i in range(8)]
    run_steps(m, bts, 20)
    models[L] = (m, bts)
res = {L: ([], []) for L in layers}
for rnd in range(3):
    for L in layers:
        m, bts = models[L]
        for _ in range(4):
            run_steps(m, bts, 5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(m, bts, 20)
            torch.cuda.synchronize()
            res[L][0].append(B * 20 / (time.perf_counter() - t0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(m, bts, 300)
        torch.cuda.synchronize()
        res[L][1].append(B * 300 / (time.perf_counter() - t0))
        print("round %d gate layer %3d: 20-step %s | steady %.0f" % (rnd, L, " ".join("%.0f" % v for v in res[L][0][-4:]), res[L][1][-1]), flush=True)
print()
for L in layers:
    print("gate layer %3d: 20-step sample median %.0f (min %.0f max %.0f) | steady state median %.0f frames/s" %
          (L, statistics.median(res[L][0]), min(res[L][0]), max(res[L][0]), statistics.median(res[L][1])))
