"""Candidate counts per frame over the bench pool and NMS time per 32-frame batch."""
import sys
import numpy as np
sys.path.insert(0, ".")
from infercam_onnx_amd import nn, synth
W, H = 640, 480
w = synth.synthetic_weights()
p = synth.gen_priors(W, H)
m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=w, priors=p, max_batch=32, profile=True, det_cap=17640)
jp = synth.synth_jpeg_pool(0, 256, W, H, quality=90, subsampling="4:2:0")
nd = []
for b in range(8):
    batch = jp[b * 32:(b + 1) * 32]
    m.profile_reset()
    for _ in range(3):
        t = m.submit_jpeg_batch(batch)
        res, st = m.wait(t)
    q = {x["name"]: x for x in m.profile_read() if x["launches"] > 0}
    nms = q["sort_nms"]["total_ms"] / q["sort_nms"]["launches"] * 1e3
    print("batch %d: detections max %d median %d; sort_nms %.1f us" % (b, max(len(r) for r in res), int(np.median([len(r) for r in res])), nms), flush=True)
