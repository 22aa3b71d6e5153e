"""NMS time against the number of candidates of the heaviest frame (scattered boxes: nearly all are
selected), with the matrix path from UFD_NMS_MAT_MIN candidates on.  Design aid.
Usage on the GPU box: UFD_NMS_MAT_MIN=<n> python tools/nms_probe.py"""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from infercam_onnx_amd import nn, synth

W, H = 640, 480
w = synth.synthetic_weights()
p = synth.gen_priors(W, H)
m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=w, priors=p, max_batch=4, profile=True, det_cap=17640)
K = m.num_priors
rng = np.random.default_rng(1)
print("UFD_NMS_MAT_MIN", os.environ.get("UFD_NMS_MAT_MIN"), "UFD_NO_NMS_MATRIX", os.environ.get("UFD_NO_NMS_MATRIX"))
for n in (32, 64, 128, 192, 256, 320, 400, 512, 768, 1024, 1536, 2048, 3000):
    conf = rng.random(K).astype(np.float32) * 0.5
    conf[rng.permutation(K)[:n]] = 0.5 + (1 + rng.random(n).astype(np.float32)) * 0.249
    c = rng.random((K, 2)).astype(np.float32)
    s = rng.random((K, 2)).astype(np.float32) * 0.03 + 0.01
    boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    scores = np.stack([1 - conf, conf], 1).astype(np.float32)
    m.profile_reset()
    for _ in range(5):
        d = m.debug_postproc(scores[None], boxes[None])
    st = {q["name"]: q for q in m.profile_read() if q["launches"] > 0}
    t = st["sort_nms"]["total_ms"] / st["sort_nms"]["launches"] * 1e3
    print("%5d candidates, %5d selected: %.1f us" % (n, len(d[0]), t), flush=True)
