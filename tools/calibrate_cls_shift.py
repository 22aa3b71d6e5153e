"""Calibrates synth.CLS_BIAS_SHIFT: per cls head, the shift that leaves ~0.5 % of priors above
confidence 0.5 on synthetic frames (uses the CPU oracle; run from the repo root)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from infercam_onnx_amd import synth, topology as T

W, H = 640, 480
blob = synth.synthetic_weights(cls_bias_shift=None)
pri = synth.gen_priors(W, H)
d = [[] for _ in range(4)]
for idx in range(6):
    x = oracle.normalize_nchw(synth.synth_frame(synth.DEFAULT_FRAME_SEED, idx, W, H))
    _, _, outs = oracle.forward(x, blob, pri, layers=True)
    for h in range(4):
        c = outs[T.CLS_LAYERS[h]]
        A = T.NUM_ANCHORS[h]
        c = c.reshape(A, 2, -1)
        d[h].append((c[:, 1] - c[:, 0]).ravel())
for h in range(4):
    v = np.concatenate(d[h])
    print(h, "n", v.size, "mean %.3f std %.3f" % (v.mean(), v.std()), "p99.5 %.3f p99 %.3f" % (np.percentile(v, 99.5), np.percentile(v, 99)))
