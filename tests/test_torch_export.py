"""Pins get_model (nn.rs:143-175: ONNX load, BN folding, priors) and the kernels' topology against
something this repo did not write: UltraFace-RFB defined as a torch.nn.Module with upstream's
structure (tools/ultraface_torch.py), exported by torch's own ONNX exporter (real exporter node
order and names, Conv + BatchNormalization pairs, priors as an embedded constant, the softmax /
box-decode tail in the graph), loaded by csrc/onnx_loader.cpp, and compared with torch's own
forward.  (The zoo file itself is a run-time download, nn.rs:21-22, and not available offline.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def exported(tmp_path_factory):
    import ultraface_torch as U

    out = {}
    d = tmp_path_factory.mktemp("onnx")
    for variant, size in ((320, (320, 240)), (640, (640, 480))):
        model = U.build_seeded(size)
        for fold in (False, True):
            path = str(d / ("ultraface-RFB-%d-%s.onnx" % (variant, "folded" if fold else "bn")))
            data = U.export_onnx(model, size, path=path, fold_bn=fold)
            out[(variant, fold)] = (path, data)
        out[(variant, "model")] = model
    return out


@pytest.mark.parametrize("variant", [320, 640])
def test_exporter_output_has_the_expected_graph(exported, variant):
    import ultraface_torch as U

    hist_bn = U.onnx_op_histogram(exported[(variant, False)][1])
    hist_fold = U.onnx_op_histogram(exported[(variant, True)][1])
    assert hist_bn["Conv"] == 52 and hist_bn["BatchNormalization"] == 35 and hist_bn["Softmax"] == 1
    assert hist_fold["Conv"] == 52 and "BatchNormalization" not in hist_fold


@pytest.mark.parametrize("variant", [320, 640])
@pytest.mark.parametrize("fold", [False, True])
def test_loader_reproduces_blob_and_priors_from_torch_export(exported, variant, fold):
    """Both export forms (Conv + BatchNormalization pairs as in the zoo file; exporter-folded):
    the loader's packed blob equals the float64-folded module parameters to f32 rounding, and the
    priors it finds in the graph equal upstream generate_priors and the library's own generator."""
    import ultraface_torch as U
    from infercam_onnx_amd import nn, synth

    path, _ = exported[(variant, fold)]
    model = exported[(variant, "model")]
    blob, pri = nn.load_onnx(path, variant)
    want = U.folded_blob(model)
    assert blob.shape == want.shape
    # (the loader folds in f32, `want` in f64: a few ulps, more where beta and the scaled mean cancel)
    assert np.allclose(blob, want, rtol=1e-5, atol=1e-6), np.abs(blob - want).max()
    assert pri is not None, "priors constant not found in the exported graph"
    W, H = (640, 480) if variant == 640 else (320, 240)
    up = model.priors.numpy()
    assert pri.shape == up.shape and np.array_equal(pri, up)
    assert np.array_equal(pri, synth.gen_priors(W, H))


@pytest.mark.parametrize("variant", [320, 640])
def test_oracle_forward_matches_torch_forward(exported, oracle_lib, variant):
    """The CPU oracle's topology (oracle/ultraface_oracle.c, restated from SURVEY 8.1) against the
    torch module's own forward (BatchNorm un-folded, float64): scores / boxes <= 1e-5."""
    import torch
    import ultraface_torch as U
    from infercam_onnx_amd import synth

    W, H = (640, 480) if variant == 640 else (320, 240)
    model = exported[(variant, "model")]
    blob = U.folded_blob(model)
    pri = model.priors.numpy()
    x = oracle_lib.normalize_nchw(synth.synth_frame(123, variant, W, H))
    with torch.no_grad():
        m64 = U.build_seeded((W, H)).double()
        ts, tb = m64(torch.from_numpy(x[None]).double())
    s, b = oracle_lib.forward(x, blob, pri)
    assert np.abs(s - ts[0].numpy()).max() <= 1e-5
    assert np.abs(b - tb[0].numpy()).max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [320, 640])
def test_gpu_from_torch_exported_onnx_matches_torch_forward(exported, variant):
    """ufd_create(weights_path = the torch-exported .onnx): the GPU's raw scores / boxes equal torch's
    own forward of the un-folded module (float64) within 1e-5 -- loader, BN folding, priors,
    topology, softmax and box decode in one check against an independent implementation."""
    import torch
    import ultraface_torch as U
    from infercam_onnx_amd import nn, synth

    W, H = (640, 480) if variant == 640 else (320, 240)
    path, _ = exported[(variant, False)]
    v = nn.UltrafaceVariant.W640H480 if variant == 640 else nn.UltrafaceVariant.W320H240
    x = np.stack([np.transpose((synth.synth_frame(124, i, W, H).astype(np.float32) / 255.0 - np.float32([0.485, 0.456, 0.406])) /
                               np.float32([0.229, 0.224, 0.225]), (2, 0, 1)) for i in range(3)]).astype(np.float32)
    with torch.no_grad():
        ts, tb = U.build_seeded((W, H)).double()(torch.from_numpy(x).double())
    with nn.UltrafaceModel(v, 0.5, 0.5, weights_path=path, max_batch=3) as m:
        s, b = m.debug_forward(x)
    assert np.abs(s - ts.numpy()).max() <= 1e-5
    assert np.abs(b - tb.numpy()).max() <= 1e-5
