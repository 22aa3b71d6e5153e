"""NMS kernel time against the number of candidates (one frame per launch).  Design aid."""
import sys
import numpy as np
sys.path.insert(0, ".")
import oracle
from infercam_onnx_amd import nn, synth

oracle.build()
W, H = 640, 480
w = synth.synthetic_weights()
p = synth.gen_priors(W, H)
m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=w, priors=p, max_batch=4, profile=True, det_cap=17640)
jp = synth.synth_jpeg_pool(0, 256, W, H, quality=90, subsampling="4:2:0")
for idx in (0, 40, 70, 100, 130, 160, 200, 220):
    x = oracle.normalize_nchw(oracle.jpeg_decode_rgb(jp[idx]))[None]
    s, b = m.debug_forward(x)
    ncand = int((s[0, :, 1] > 0.5).sum())
    m.profile_reset()
    for _ in range(5):
        d = m.debug_postproc(s, b)
    st = {q["name"]: q for q in m.profile_read()}
    t = st["sort_nms"]["total_ms"] / max(st["sort_nms"]["launches"], 1) * 1e3
    print("frame %3d: %5d candidates, %4d selected, sort_nms %.1f us" % (idx, ncand, len(d[0]), t), flush=True)
