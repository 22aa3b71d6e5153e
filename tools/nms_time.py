"""Device time of the post-processing launches (k_sort_nms + k_nms_matrix + k_nms_scan) for ONE frame with n candidates
(random boxes of the given size), alone on the GPU: python tools/nms_time.py [box size]"""
import sys

import numpy as np

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infercam_onnx_amd import nn, synth

size = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03
W, H = 640, 480
m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=synth.synthetic_weights(), priors=synth.gen_priors(W, H),
                      max_batch=4, profile=True, det_cap=17640)
K = m.num_priors
rng = np.random.default_rng(3)
for n in (32, 256, 257, 1024, 2048, 2049, 3000, 4096, 4097, 6000, 8000, 8192, 8193, 12000):
    conf = rng.random(K).astype(np.float32) * 0.5
    hot = rng.permutation(K)[:n]
    conf[hot] = 0.5 + (1 + rng.random(n).astype(np.float32)) * 0.249
    c = rng.random((K, 2)).astype(np.float32)
    s = (rng.random((K, 2)).astype(np.float32) * size + 0.005)
    boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    scores = np.stack([1 - conf, conf], 1).astype(np.float32)
    m.debug_postproc(scores[None], boxes[None])
    m.profile_reset()
    for _ in range(5):
        res = m.debug_postproc(scores[None], boxes[None])
    q = {x["name"]: x for x in m.profile_read() if x["launches"] > 0}
    print("n = %5d: selected %5d, sort + NMS %.1f us" % (n, len(res[0]), q["sort_nms"]["total_ms"] / q["sort_nms"]["launches"] * 1e3), flush=True)
m.close()
