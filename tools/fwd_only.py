"""Runs only the CNN forward (ufd_debug_forward) a few times: a small target for rocprofv3 --pmc."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from infercam_onnx_amd import nn, synth
W, H, B = 640, 480, 32
m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, max_batch=B, weights=synth.synthetic_weights(),
                      priors=synth.gen_priors(W, H), max_src=(W, H))
x = np.random.default_rng(0).standard_normal((B, 3, H, W)).astype(np.float32)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    m.debug_forward(x)
m.close()
