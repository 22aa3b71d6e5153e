"""Two (three) handles on ONE GPU, each fed by its own thread with six batches in flight -- a server that runs both model
variants, or two schedulers in one process: their streams share the runtime's four hardware queues.
    python tools/two_handles.py [mixed]      (mixed: the second handle is UltraFace-320 on 320x240 frames)"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infercam_onnx_amd import nn, synth

B, STEPS = 32, 300
weights = synth.synthetic_weights()
mixed = "mixed" in sys.argv[1:]


def make(k):
    v = nn.UltrafaceVariant.W320H240 if (mixed and k % 2) else nn.UltrafaceVariant.W640H480
    W, H = v.width_height()
    m = nn.UltrafaceModel(v, 0.5, 0.5, max_batch=B, weights=weights, priors=synth.gen_priors(W, H), max_src=(W, H), det_cap=256)
    jp = synth.synth_jpeg_pool(k, 128, W, H)
    return m, [m._prep_batch(jp[i * B:(i + 1) * B]) for i in range(4)]


def drive(m, bs, steps, out, k):
    infl = []
    t0 = time.perf_counter()
    for s in range(steps):
        if len(infl) >= 6:
            m.wait(infl.pop(0), collect=False)
        infl.append(m.submit_jpeg_batch(bs[s % 4]))
    for t in infl:
        m.wait(t, collect=False)
    out[k] = steps * B / (time.perf_counter() - t0)


for nh in (1, 2, 3):
    hs = [make(k) for k in range(nh)]
    for m, bs in hs:
        drive(m, bs, 12, {}, 0)
    out = {}
    th = [threading.Thread(target=drive, args=(m, bs, STEPS, out, k)) for k, (m, bs) in enumerate(hs)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    print("handles %d%s: total %.0f frames/s (%s)" % (nh, " mixed 640/320" if mixed else "", nh * STEPS * B / el,
                                                      ", ".join("%.0f" % out[k] for k in range(nh))), flush=True)
    for m, _ in hs:
        m.close()
