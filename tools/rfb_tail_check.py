"""Where k_rfb_tail's output differs from the two-launch form, per batch size (debugging aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from infercam_onnx_amd import nn, synth

W, H = 640, 480
weights = synth.synthetic_weights()
pri = synth.gen_priors(W, H)
for B in (1, 2, 4, 8):
    x = np.stack([oracle.normalize_nchw(synth.synth_frame(94, i % 7, W, H)) for i in range(B)])
    kw = dict(weights=weights, priors=pri, max_batch=B, tap_layers=True)
    a = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, extra_flags=nn.UFD_FLAG_NO_RFB_TAIL, **kw)
    b = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, **kw)
    sa, ba = a.debug_forward(x)
    sb, bb = b.debug_forward(x)
    print("B=%d scores diff %.3g boxes diff %.3g" % (B, np.abs(sa - sb).max(), np.abs(ba - bb).max()), flush=True)
    for f in range(B):
        for layer in (24, 30, 34, 42):
            try:
                ra, rb = a.debug_layer_output(layer, f), b.debug_layer_output(layer, f)
            except nn.UfdError as e:
                print("  layer", layer, "absent", e)
                continue
            d = np.abs(ra - rb)
            if d.max() > 1e-4 * max(np.abs(ra).max(), 1e-6):
                idx = np.argwhere(d.reshape(ra.shape) > 1e-4 * np.abs(ra).max())
                print("  B=%d frame %d layer %d: max diff %.3g at %d positions, first %s last %s shape %s" % (B, f, layer, d.max(), len(idx), idx[0], idx[-1], ra.shape))
    a.close(); b.close()
