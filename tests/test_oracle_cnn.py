"""Oracle pinning, rows A5/A6: the C forward pass against torch conv2d goldens (float64 torch run
committed as tests/golden/cnn_320.npz by tools/make_golden.py) and a live torch cross-check."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


def test_topology_totals(oracle_lib):
    from infercam_onnx_amd import topology as T

    assert oracle_lib.weight_floats() == T.total_weight_floats() == 273888  # 1 095 552 B
    assert oracle_lib.num_priors(640, 480) == T.num_priors(640, 480) == 17640
    assert oracle_lib.num_priors(320, 240) == T.num_priors(320, 240) == 4420
    assert T.macs(640, 480) == 399157760 and T.macs(320, 240) == 100418560
    for i, (s, o) in enumerate(zip(T.CONVS, oracle_lib.conv_specs())):
        assert (s.cin, s.cout, s.k, s.stride, s.pad, s.dil, s.groups, s.relu) == tuple(o.values()), i


def test_priors_match_upstream_generator(oracle_lib):
    from infercam_onnx_amd import synth

    for w, h in ((640, 480), (320, 240)):
        p = oracle_lib.gen_priors(w, h)
        assert np.array_equal(p, synth.gen_priors(w, h))
        assert p.min() >= 0 and p.max() <= 1
    p = oracle_lib.gen_priors(640, 480)
    assert np.allclose(p[0], [0.5 / 80, 0.5 / 60, 10 / 640, 10 / 480])
    assert np.allclose(p[-1], [9.5 / 10, 7.5 / 8, 256 / 640, 256 / 480])


def test_forward_matches_torch_golden(oracle_lib, weights):
    from infercam_onnx_amd import synth

    z = np.load(os.path.join(G, "cnn_320.npz"))
    assert np.array_equal(z["frame"], synth.synth_frame(synth.DEFAULT_FRAME_SEED, 0, 320, 240))
    x = oracle_lib.normalize_nchw(z["frame"])
    scores, boxes, outs = oracle_lib.forward(x, weights, synth.gen_priors(320, 240), layers=True)
    assert np.abs(scores - z["scores"]).max() < 2e-5
    assert np.abs(boxes - z["boxes"]).max() < 2e-5
    for i, o in enumerate(outs):
        assert abs(np.abs(o).max() - z["layer_absmax"][i]) <= 1e-4 * max(1.0, z["layer_absmax"][i]), i
        assert abs(o.mean() - z["layer_mean"][i]) <= 1e-4 * max(1.0, abs(z["layer_mean"][i])), i


@pytest.mark.parametrize("blob", ["he_normal", "bn_folded_like"])
def test_forward_per_layer_against_live_torch(oracle_lib, weights, blob):
    """(bn_folded_like: a scale of its own per output channel over a ratio of 16 and large biases -- the value ranges of a
    BatchNorm-folded checkpoint such as the reference's zoo file, which cannot be had offline: synth.bn_folded_like_weights)"""
    import torch
    import torch.nn.functional as F
    from infercam_onnx_amd import synth, topology as T

    if blob == "bn_folded_like":
        weights = synth.bn_folded_like_weights()
    torch.set_num_threads(4)
    W, H = 320, 240
    x = oracle_lib.normalize_nchw(synth.synth_frame(5, 5, W, H))
    _, _, outs = oracle_lib.forward(x, weights, synth.gen_priors(W, H), layers=True)
    acts = {}
    xin = torch.from_numpy(x)[None].double()
    for i, s in enumerate(T.CONVS):
        inp = xin if s.src == -1 else (torch.cat([acts[15], acts[18], acts[22]], 1) if s.src == -2 else acts[s.src])
        w, b = synth.layer_params(weights, i)
        y = F.conv2d(inp, torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=s.stride, padding=s.pad,
                     dilation=s.dil, groups=s.groups)
        y = torch.relu(acts[23] + y) if i == 24 else (torch.relu(y) if s.relu else y)
        acts[i] = y
        ref = y[0].numpy()
        assert np.abs(outs[i] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0), "layer %d %s" % (i, s.name)


def test_reference_boxes_if_pinned(oracle_lib):
    """tools/pin_reference.py writes tests/golden/reference_boxes.npz the day the zoo files (nn.rs:21-22) can be had: the
    oracle's detections on the reference's eight pictures with the REAL weights, after asserting the reference's own face
    counts (integration_tests.rs:20-29).  With that file and the .onnx in the cache path, every later oracle build must
    reproduce them; without them the test reports that the pin is still open."""
    import json
    from helpers import REFERENCE_PINS
    from infercam_onnx_amd import nn, synth

    G = os.path.join(os.path.dirname(__file__), "golden")
    npz = os.path.join(G, "reference_boxes.npz")
    if not os.path.exists(npz):
        REFERENCE_PINS["reference_boxes"] = "NOT CHECKED (tests/golden/reference_boxes.npz not written yet: tools/pin_reference.py)"
        pytest.skip("no zoo weights have been seen yet")
    z = np.load(npz)
    meta = json.load(open(os.path.join(G, "test_pics.json")))
    checked = 0
    for variant, (W, H) in ((640, (640, 480)), (320, (320, 240))):
        path = os.path.join(os.environ.get("XDG_CACHE_HOME", os.path.expanduser("~/.cache")), "infercam_onnx", "ultraface-RFB-%d.onnx" % variant)
        if "sha256_%d" % variant not in z.files or not os.path.exists(path):
            continue
        w, pri = nn.load_onnx(path, variant)
        pri = pri if pri is not None else synth.gen_priors(W, H)
        for f, info in meta.items():
            got = np.asarray(oracle_lib.infer_jpeg(open(os.path.join(G, "test_pics", f), "rb").read(), W, H, w, pri, 0.5, 0.5), np.float32).reshape(-1, 5)
            if variant == 640:
                assert len(got) == info["reference_face_count"], f
            assert got.shape == z["%d/%s" % (variant, f)].shape and np.allclose(got, z["%d/%s" % (variant, f)], atol=1e-6), f
            checked += 1
    if not checked:
        pytest.skip("reference_boxes.npz is there but the .onnx files are not in the cache path")
    REFERENCE_PINS["reference_boxes"] = "checked: %d pictures" % checked
