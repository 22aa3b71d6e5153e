"""Entropy-decoder time on FLAT content (a lens cap, a night frame, an over-exposed wall): blocks of two short symbols in a
periodic stream, where a decoder started out of step does not fall back into step, so the speculation of the
self-synchronising decoder does not find the true chain and k_huff_resolve walks it.
    python tools/flat_frame_time.py
prints, per batch composition, the wall time of a batch of 32 alone on the GPU and the per-kernel device time of the chain."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infercam_onnx_amd import nn, synth

W, H, B = 640, 480, 32
m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=synth.synthetic_weights(), priors=synth.gen_priors(W, H),
                      max_batch=B, profile=True, device_entropy=True)
normal = [synth.encode_jpeg(synth.synth_frame(77, i, W, H), subsampling="4:2:0", quality=90) for i in range(B)]
black = synth.encode_jpeg(np.zeros((H, W, 3), np.uint8), subsampling="4:2:0", quality=90)
gray = synth.encode_jpeg(np.full((H, W, 3), 117, np.uint8), subsampling="4:2:0", quality=90)
half = np.array(synth.synth_frame(77, 3, W, H))
half[H // 2:] = 16  # lower half of the picture flat
half = synth.encode_jpeg(half, subsampling="4:2:0", quality=90)
cases = [("32 camera-like frames", normal), ("31 + one black frame", normal[:31] + [black]), ("31 + one half-flat frame", normal[:31] + [half]),
         ("32 black frames", [black] * B), ("32 flat gray frames", [gray] * B), ("one black frame alone", [black]), ("one camera-like frame alone", normal[:1])]
print("JPEG bytes: camera-like %d, black %d, gray %d, half-flat %d" % (len(normal[0]), len(black), len(gray), len(half)))
for name, batch in cases:
    for _ in range(3):
        m.infer_jpeg_batch(batch)
    m.profile_reset()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        dets, st = m.infer_jpeg_batch(batch)
    el = (time.perf_counter() - t0) / n
    assert all(s == 0 for s in st), st
    q = {x["name"].split(":")[0]: x for x in m.profile_read() if x["launches"] > 0}
    chain = ["huff_unstuff", "huff_seed", "huff_extend", "huff_link", "huff_resolve", "huff_write", "dc_prefix"]
    parts = " ".join("%s %.0f" % (k.replace("huff_", ""), q[k]["total_ms"] / n * 1e3) for k in chain if k in q)
    print("%-28s batch wall %.3f ms; chain us per batch: %s" % (name, el * 1e3, parts), flush=True)
m.close()
