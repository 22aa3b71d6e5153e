"""GPU parity tests: every stage of the hot path through the C ABI against the CPU oracle."""
import os

import numpy as np
import pytest

from helpers import assert_dets_match, dets_array

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANT_WH = {640: (640, 480), 320: (320, 240)}


def make_model(variant, weights, **kw):
    from infercam_onnx_amd import nn, synth

    v = nn.UltrafaceVariant.W640H480 if variant == 640 else nn.UltrafaceVariant.W320H240
    W, H = v.width_height()
    kw.setdefault("max_src", (1280, 960))
    kw.setdefault("det_cap", 17640)
    return nn.UltrafaceModel(v, 0.5, 0.5, weights=weights, priors=synth.gen_priors(W, H), **kw)


@pytest.fixture(scope="module")
def model320(weights):
    m = make_model(320, weights, max_batch=4, keep_layers=True, host_entropy=True)  # Huffman on host workers
    yield m
    m.close()


@pytest.fixture(scope="module")
def model640(weights):
    m = make_model(640, weights, max_batch=8, host_entropy=True)
    yield m
    m.close()


# ---------------------------------------------------------------- A1
@pytest.mark.parametrize("size", [(640, 480), (320, 240), (641, 479), (17, 9), (1, 1), (1280, 720), (100, 37)])
@pytest.mark.parametrize("subsampling", ["4:2:0", "4:2:2", "4:4:4"])
def test_jpeg_decode_bit_exact(model320, oracle_lib, size, subsampling):
    from infercam_onnx_amd import synth

    w, h = size
    for kw in ({}, {"restart_rows": 1}, {"progressive": True}):
        jpeg = synth.encode_jpeg(synth.synth_frame(11, w + h, w, h), quality=90, subsampling=subsampling, **kw)
        got = model320.debug_decode_jpeg(jpeg)
        ref = oracle_lib.jpeg_decode_rgb(jpeg)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), "decode mismatch %s %s %s: %d px differ" % (size, subsampling, kw, (got != ref).sum())


def test_jpeg_decode_gray_and_quality(model320, oracle_lib):
    from PIL import Image
    import io
    from infercam_onnx_amd import synth

    f = synth.synth_frame(5, 0, 333, 111)
    bio = io.BytesIO()
    Image.fromarray(f[:, :, 0]).save(bio, "JPEG", quality=85)
    jpeg = bio.getvalue()
    assert np.array_equal(model320.debug_decode_jpeg(jpeg), oracle_lib.jpeg_decode_rgb(jpeg))
    for q in (30, 100):
        jpeg = synth.encode_jpeg(f, quality=q)
        assert np.array_equal(model320.debug_decode_jpeg(jpeg), oracle_lib.jpeg_decode_rgb(jpeg))


def test_corrupt_jpeg_is_an_error_not_a_crash(model320):
    from infercam_onnx_amd import nn, synth

    jpeg = synth.encode_jpeg(synth.synth_frame(1, 1, 320, 240))
    for bad in (jpeg[:500], jpeg[:-2], b"\xff\xd8\xff", b"not a jpeg"):
        with pytest.raises(nn.UfdError) as e:
            model320.infer_jpeg(bad)
        assert e.value.code == nn.UFD_E_DECODE
    res, status = model320.infer_jpeg_batch([jpeg, jpeg[:500], jpeg])
    assert status[0] == 0 and status[2] == 0 and status[1] == nn.UFD_E_DECODE
    assert res[1] is None and res[0] == res[2]


def test_create_refuses_a_batch_whose_tensors_outgrow_32_bit_offsets(weights):
    """The convolution kernels address a tensor by a wave-uniform base + 32-bit per-lane byte offsets: a max_batch whose
    largest activation tensor (the 640 stem's 16 x 240 x 320 floats per frame) would reach 4 GiB is refused at ufd_create."""
    from infercam_onnx_amd import nn, synth

    with pytest.raises(nn.UfdError) as e:
        nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=weights, priors=synth.gen_priors(640, 480), max_batch=1000)
    assert e.value.code == nn.UFD_E_TOO_LARGE and "4 GiB" in str(e.value)


# ---------------------------------------------------------------- A2-A4
@pytest.mark.parametrize("src", [(1280, 720), (640, 427), (640, 960), (320, 240), (640, 480), (100, 37), (333, 500),
                                 (641, 479), (1279, 719), (17, 9), (1, 1), (1280, 960), (3, 700)])
@pytest.mark.parametrize("variant", [320, 640])
def test_preproc_bit_exact(model320, model640, oracle_lib, src, variant):
    from infercam_onnx_amd import synth

    m = model320 if variant == 320 else model640
    W, H = VARIANT_WH[variant]
    rgb = synth.synth_frame(7, src[0], src[0], src[1])
    got = m.debug_preproc(rgb)
    ref = oracle_lib.normalize_nchw(oracle_lib.resize_triangle(rgb, W, H))
    assert np.array_equal(got, ref), "preproc mismatch: %d values differ, max %g" % ((got != ref).sum(), np.abs(got - ref).max())


# ---------------------------------------------------------------- A6
def test_forward_per_layer(model320, oracle_lib, weights):
    from infercam_onnx_amd import synth

    W, H = 320, 240
    pri = synth.gen_priors(W, H)
    x = np.stack([oracle_lib.normalize_nchw(synth.synth_frame(21, i, W, H)) for i in range(2)])
    scores, boxes = model320.debug_forward(x)
    for f in range(2):
        rs, rb, outs = oracle_lib.forward(x[f], weights, pri, layers=True)
        for li, ref in enumerate(outs):
            got = model320.debug_layer_output(li, f).reshape(ref.shape)
            scale = max(np.abs(ref).max(), 1e-6)
            err = np.abs(got - ref).max() / scale
            assert err <= 1e-5, "layer %d frame %d: rel err %g" % (li, f, err)
        assert np.abs(scores[f] - rs).max() <= 1e-5
        assert np.abs(boxes[f] - rb).max() <= 1e-5


def test_forward_640_batch(model640, oracle_lib, weights):
    from infercam_onnx_amd import synth

    W, H = 640, 480
    pri = synth.gen_priors(W, H)
    x = np.stack([oracle_lib.normalize_nchw(synth.synth_frame(22, i, W, H)) for i in range(3)])
    scores, boxes = model640.debug_forward(x)
    for f in range(3):
        rs, rb = oracle_lib.forward(x[f], weights, pri)
        assert np.abs(scores[f] - rs).max() <= 1e-5
        assert np.abs(boxes[f] - rb).max() <= 1e-5


# ---------------------------------------------------------------- A7-A10
def test_postproc_exact_on_same_inputs(model320, oracle_lib):
    rng = np.random.default_rng(5)
    K = model320.num_priors
    for trial, frac in enumerate((0.0, 0.002, 0.05, 0.6, 1.0)):
        conf = rng.random(K).astype(np.float32) * 0.5
        hot = rng.random(K) < frac
        conf[hot] = 0.5 + rng.random(hot.sum()).astype(np.float32) * 0.5
        if trial == 2:
            conf[hot] = np.round(conf[hot] * 8) / 8  # exact ties: processing order = higher index first
        scores = np.stack([1 - conf, conf], 1).astype(np.float32)
        c = rng.random((K, 2)).astype(np.float32)
        s = (rng.random((K, 2)).astype(np.float32) * 0.2 + 0.01)
        boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
        boxes[::97, 2] = boxes[::97, 0] - 0.01  # degenerate boxes: zero area
        got = dets_array(model320.debug_postproc(scores[None], boxes[None])[0])
        ref = oracle_lib.postproc(scores, boxes, 0.5, 0.5)
        assert got.shape == ref.shape, "trial %d: %d vs %d" % (trial, len(got), len(ref))
        assert np.array_equal(got, ref), "trial %d differs" % trial


@pytest.mark.parametrize("size", [0.02, 0.12, 0.5])
def test_postproc_many_candidates_matrix_path(model320, oracle_lib, size):
    """Frames with 257..2048 candidates leave k_sort_nms after the sort (k_nms_matrix + k_nms_scan);
    lighter and heavier frames of the same batch stay inside it.  Bit-exact selection and order
    for sparse, dense and mostly-overlapping boxes, confidence ties and zero-area boxes."""
    rng = np.random.default_rng(int(size * 1000))
    K = model320.num_priors
    counts = [257, 2048, 64, 2049, 300, 1000, 1984, 0, 511, 1360, 256, 2047]
    for b0 in range(0, len(counts), 4):
        sc, bx = [], []
        for f, n in enumerate(counts[b0:b0 + 4]):
            conf = rng.random(K).astype(np.float32) * 0.5
            hot = rng.permutation(K)[:n]
            conf[hot] = 0.5 + (1 + rng.random(n).astype(np.float32)) * 0.249
            if f % 2 == 1:
                conf[hot] = np.round(conf[hot] * 64) / 64  # many exact ties
            c = rng.random((K, 2)).astype(np.float32)
            s = (rng.random((K, 2)).astype(np.float32) * size + 0.01)
            boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
            boxes[::53, 2] = boxes[::53, 0] - 0.01  # degenerate boxes: zero area
            sc.append(np.stack([1 - conf, conf], 1).astype(np.float32))
            bx.append(boxes)
        got = model320.debug_postproc(np.stack(sc), np.stack(bx))
        for f, n in enumerate(counts[b0:b0 + 4]):
            ref = oracle_lib.postproc(sc[f], bx[f], 0.5, 0.5)
            g = dets_array(got[f])
            assert int((sc[f][:, 1] > 0.5).sum()) == n
            assert g.shape == ref.shape, "%d candidates: %d vs %d selected" % (n, len(g), len(ref))
            assert np.array_equal(g, ref), "%d candidates differ" % n


@pytest.mark.parametrize("size", [0.015, 0.08])
def test_postproc_two_matrix_halves(model640, oracle_lib, size):
    """Frames with 2049..4096 candidates (nn.rs:198-224 has no cap): sorted in LDS like the lighter ones, suppression
    matrix in four quadrants, scan in two halves with the off-diagonal square folded in between -- no HBM sort, no
    one-CU block loop.  Beyond 4096 the in-kernel block loop still answers.  Bit-exact selection and order next to
    light frames in the same batch, with confidence ties and zero-area boxes."""
    rng = np.random.default_rng(int(size * 1000) + 7)
    K = model640.num_priors
    counts = [2049, 3000, 100, 4096, 4097, 2048, 3583, 0, 2112, 6000, 4095, 17640]
    for b0 in range(0, len(counts), 4):
        sc, bx = [], []
        for f, n in enumerate(counts[b0:b0 + 4]):
            conf = rng.random(K).astype(np.float32) * 0.5
            hot = rng.permutation(K)[:n]
            conf[hot] = 0.5 + (1 + rng.random(n).astype(np.float32)) * 0.249
            if f % 2 == 1:
                conf[hot] = np.round(conf[hot] * 256) / 256  # many exact ties
            c = rng.random((K, 2)).astype(np.float32)
            s = (rng.random((K, 2)).astype(np.float32) * size + 0.005)
            boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
            boxes[::53, 2] = boxes[::53, 0] - 0.01  # degenerate boxes: zero area
            sc.append(np.stack([1 - conf, conf], 1).astype(np.float32))
            bx.append(boxes)
        got = model640.debug_postproc(np.stack(sc), np.stack(bx))
        for f, n in enumerate(counts[b0:b0 + 4]):
            ref = oracle_lib.postproc(sc[f], bx[f], 0.5, 0.5)
            g = dets_array(got[f])
            assert int((sc[f][:, 1] > 0.5).sum()) == n
            assert g.shape == ref.shape, "%d candidates: %d vs %d selected" % (n, len(g), len(ref))
            assert np.array_equal(g, ref), "%d candidates differ" % n


@pytest.mark.parametrize("size", [0.012, 0.06])
def test_postproc_up_to_four_matrix_squares(model640, oracle_lib, size):
    """Frames with 4097..8192 candidates (round 4: the matrix path's cap moved from 4096 to 8192; the reference has none,
    nn.rs:198-224): keys sorted as eight 1024-key blocks in two LDS passes and merged by rank, the suppression matrix in up to
    ten 2048 x 2048 squares, the scan over up to four diagonal squares with the off-diagonal ones folded in between.  Both
    sides of every boundary (4096/4097, 6144/6145, 8192/8193: the last falls back to the in-kernel block loop), next to light
    frames in the same batch, with confidence ties and zero-area boxes: bit-exact selection and order."""
    rng = np.random.default_rng(int(size * 1000) + 19)
    K = model640.num_priors
    counts = [4097, 8192, 300, 6144, 6145, 5000, 8193, 0, 8191, 7000, 4096, 9000]
    for b0 in range(0, len(counts), 4):
        sc, bx = [], []
        for f, n in enumerate(counts[b0:b0 + 4]):
            conf = rng.random(K).astype(np.float32) * 0.5
            hot = rng.permutation(K)[:n]
            conf[hot] = 0.5 + (1 + rng.random(n).astype(np.float32)) * 0.249
            if f % 2 == 1:
                conf[hot] = np.round(conf[hot] * 512) / 512  # many exact ties
            c = rng.random((K, 2)).astype(np.float32)
            s = (rng.random((K, 2)).astype(np.float32) * size + 0.004)
            boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
            boxes[::53, 2] = boxes[::53, 0] - 0.01  # degenerate boxes: zero area
            sc.append(np.stack([1 - conf, conf], 1).astype(np.float32))
            bx.append(boxes)
        got = model640.debug_postproc(np.stack(sc), np.stack(bx))
        for f, n in enumerate(counts[b0:b0 + 4]):
            ref = oracle_lib.postproc(sc[f], bx[f], 0.5, 0.5)
            g = dets_array(got[f])
            assert int((sc[f][:, 1] > 0.5).sum()) == n
            assert g.shape == ref.shape, "%d candidates: %d vs %d selected" % (n, len(g), len(ref))
            assert np.array_equal(g, ref), "%d candidates differ" % n


# ---------------------------------------------------------------- end to end
@pytest.mark.parametrize("variant,src", [(320, (320, 240)), (640, (640, 480)), (640, (1280, 720)), (320, (1280, 720)),
                                         (640, (640, 427))])
def test_infer_jpeg_end_to_end(model320, model640, oracle_lib, weights, variant, src):
    from infercam_onnx_amd import synth

    m = model320 if variant == 320 else model640
    W, H = VARIANT_WH[variant]
    pri = synth.gen_priors(W, H)
    jpegs = [synth.encode_jpeg(synth.synth_frame(31, i, src[0], src[1])) for i in range(3)]
    res, status = m.infer_jpeg_batch(jpegs)
    assert status == [0, 0, 0]
    for j, r in zip(jpegs, res):
        ref = oracle_lib.infer_jpeg(j, W, H, weights, pri, 0.5, 0.5)
        x = oracle_lib.normalize_nchw(oracle_lib.resize_triangle(oracle_lib.jpeg_decode_rgb(j), W, H))
        scores, _ = oracle_lib.forward(x, weights, pri)
        assert_dets_match(dets_array(r), ref, scores=scores, what="jpeg %s->%d" % (src, variant))
        assert_dets_match(dets_array(m.infer_jpeg(j)), ref, scores=scores)


def test_infer_rgb_matches_reference_run(model640, oracle_lib, weights):
    from infercam_onnx_amd import synth

    pri = synth.gen_priors(640, 480)
    for src in ((640, 480), (640, 427), (1280, 720)):
        rgb = synth.synth_frame(41, 3, src[0], src[1])
        ref = oracle_lib.infer_rgb(rgb, 640, 480, weights, pri, 0.5, 0.5)
        x = oracle_lib.normalize_nchw(oracle_lib.resize_triangle(rgb, 640, 480))
        scores, _ = oracle_lib.forward(x, weights, pri)
        assert_dets_match(dets_array(model640.run(rgb)), ref, scores=scores, what="rgb %s" % (src,))


def test_async_submit_wait_order(model640, oracle_lib, weights):
    from infercam_onnx_amd import synth

    jpegs = [synth.encode_jpeg(synth.synth_frame(51, i, 640, 480)) for i in range(8)]
    sync, _ = model640.infer_jpeg_batch(jpegs)
    t1 = model640.submit_jpeg_batch(jpegs[:4])
    t2 = model640.submit_jpeg_batch(jpegs[4:])
    r2, _ = model640.wait(t2)
    r1, _ = model640.wait(t1)
    # (kernel selection, e.g. split-K, depends on the batch size: equal within fp32 rounding)
    for a, b in zip(r1 + r2, sync):
        assert_dets_match(dets_array(a), dets_array(b), atol=1e-5)


# ---------------------------------------------------------------- A1 with device entropy decoding
@pytest.fixture(scope="module")
def model320_dev(weights):
    m = make_model(320, weights, max_batch=4, device_entropy=True)
    yield m
    m.close()


@pytest.fixture(scope="module")
def model640_dev(weights):
    m = make_model(640, weights, max_batch=8, device_entropy=True)
    yield m
    m.close()


@pytest.mark.parametrize("size", [(640, 480), (641, 479), (1280, 720), (17, 9)])
@pytest.mark.parametrize("subsampling", ["4:2:0", "4:2:2", "4:4:4"])
def test_device_entropy_decode_bit_exact(model320_dev, oracle_lib, size, subsampling):
    """Restart-interval streams on the device entropy pipeline: same pixels as the oracle for
    every interval length."""
    from infercam_onnx_amd import synth

    w, h = size
    rgb = synth.synth_frame(12, w * 3 + h, w, h)
    for kw in ({"restart_rows": 1}, {"restart_rows": 2}, {"restart_rows": 1, "quality": 30}, {"restart_rows": 1, "optimize": True}):
        jpeg = synth.encode_jpeg(rgb, subsampling=subsampling, **kw)
        assert b"\xff\xdd" in jpeg  # DRI present
        got = model320_dev.debug_decode_jpeg(jpeg)
        assert np.array_equal(got, oracle_lib.jpeg_decode_rgb(jpeg)), (size, subsampling, kw)


@pytest.mark.parametrize("size", [(640, 480), (641, 479), (1280, 720), (17, 9), (8, 8), (320, 240)])
@pytest.mark.parametrize("subsampling", ["4:2:0", "4:2:2", "4:4:4"])
def test_device_sync_decoder_bit_exact(model320_dev, oracle_lib, size, subsampling):
    """Streams without restart markers take the self-synchronising device decoder: same pixels
    as the oracle at every quality (long and short symbols, optimised tables, noise)."""
    from infercam_onnx_amd import synth

    w, h = size
    rgb = synth.synth_frame(13, w * 5 + h, w, h)
    noise = np.random.default_rng(w * h).integers(0, 256, size=rgb.shape, dtype=np.uint8)
    for img, kw in ((rgb, {}), (rgb, {"quality": 30}), (rgb, {"quality": 100}), (rgb, {"optimize": True}),
                    (noise, {"quality": 95}), (np.zeros_like(rgb), {})):
        jpeg = synth.encode_jpeg(img, subsampling=subsampling, **kw)
        assert b"\xff\xdd" not in jpeg  # no DRI
        got = model320_dev.debug_decode_jpeg(jpeg)
        assert np.array_equal(got, oracle_lib.jpeg_decode_rgb(jpeg)), (size, subsampling, kw)


_FLOOR_SCRIPT = """
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from infercam_onnx_amd import nn, synth
w = synth.synthetic_weights()
m = nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=w, priors=synth.gen_priors(320, 240), max_batch=2,
                      max_src=(1280, 960), device_entropy=True, extra_flags=int(sys.argv[1]))
h = hashlib.sha256()
for i, (wd, ht, kw) in enumerate([(640, 480, {}), (640, 480, {"quality": 30}), (641, 479, {"quality": 100}), (320, 240, {"optimize": True}),
                                  (17, 9, {}), (1280, 720, {"quality": 60})]):
    jpeg = synth.encode_jpeg(synth.synth_frame(15, i, wd, ht), subsampling="4:2:0", **kw)
    h.update(np.ascontiguousarray(m.debug_decode_jpeg(jpeg)).tobytes())
smooth = synth.encode_jpeg(np.full((480, 640, 3), 117, np.uint8), subsampling="4:2:0")  # two-symbol blocks: the longest way back into step
h.update(np.ascontiguousarray(m.debug_decode_jpeg(smooth)).tobytes())
print("floor-hash", h.hexdigest())
"""


def test_sync_decoder_subsequence_floor_does_not_change_a_pixel():
    """The self-synchronising decoder cuts a frame's stream into subsequences of >= 64 bytes when the batch fills the GPU
    with lanes anyway, >= 32 bytes for a frame or a few at a time (model.cpp, sub_floor: 167 -> 129 us of entropy chain for
    ONE 640x480 frame).  The same frames decoded one at a time under both floors (UFD_FLAG_SUBSEQ_32 / _64 force either), each in a
    process of its own: identical pixels (against the oracle: the single-frame tests above run under the 32-byte floor, the
    batch tests and bench.py's `verified` under the 64-byte one)."""
    import subprocess
    import sys

    hashes = []
    from infercam_onnx_amd import nn

    for flag in (nn.UFD_FLAG_SUBSEQ_64, nn.UFD_FLAG_SUBSEQ_32):
        r = subprocess.run([sys.executable, "-c", _FLOOR_SCRIPT % ROOT, str(flag)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "floor-hash" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
        hashes.append(r.stdout.split("floor-hash")[1].split()[0])
    assert hashes[0] == hashes[1], hashes


def test_device_sync_decoder_grayscale_and_coefficients(model320_dev, oracle_lib):
    from PIL import Image
    import io
    from infercam_onnx_amd import synth

    rgb = synth.synth_frame(14, 3, 333, 217)
    buf = io.BytesIO()
    Image.fromarray(rgb).convert("L").save(buf, format="JPEG", quality=85)
    jpeg = buf.getvalue()
    got = model320_dev.debug_decode_jpeg(jpeg)
    assert np.array_equal(got, oracle_lib.jpeg_decode_rgb(jpeg))


@pytest.mark.parametrize("subsampling", ["4:2:0", "4:2:2"])
def test_mjpg_without_dht_on_the_device_entropy_path(weights, oracle_lib, subsampling):
    """camera MJPG may omit DHT (=> Annex-K default tables).  The DHT segments are stripped from a
    non-optimised libjpeg-turbo stream -- which carries exactly those tables -- and the bare stream goes through the
    DEVICE entropy decoder (profile: huff_write ran, no host coefficient copy); pixels and detections must equal the
    oracle's decode of the ORIGINAL stream (identical tables => identical pixels), with and without restart markers."""
    from infercam_onnx_amd import synth

    m = make_model(320, weights, max_batch=4, device_entropy=True, profile=True)
    try:
        for kw in ({}, {"restart_rows": 1}, {"quality": 50}):
            frames = [synth.synth_frame(31, i, 320, 240) for i in range(3)]
            jpegs = [synth.encode_jpeg(f, subsampling=subsampling, **kw) for f in frames]
            bare = [synth.strip_dht(j) for j in jpegs]
            assert all(b"\xff\xc4" not in b[:b.index(b"\xff\xda")] for b in bare)
            for j, b in zip(jpegs, bare):
                assert np.array_equal(m.debug_decode_jpeg(b), oracle_lib.jpeg_decode_rgb(j)), (subsampling, kw)
            m.profile_reset()
            res_bare, st = m.infer_jpeg_batch(bare)
            names = {p["name"] for p in m.profile_read()}
            assert st == [0] * 3 and "huff_write" in names and "h2d_coef" not in names, (st, names)
            res_full, st = m.infer_jpeg_batch(jpegs)
            assert st == [0] * 3
            for a, b in zip(res_bare, res_full):
                assert np.array_equal(dets_array(a), dets_array(b))
    finally:
        m.close()


def test_device_sync_decoder_is_the_path_taken(weights):
    """The profile names the kernels that ran: the sync decoder, not the host Huffman copy."""
    from infercam_onnx_amd import synth

    m = make_model(640, weights, max_batch=4, device_entropy=True, profile=True)
    try:
        jpegs = [synth.encode_jpeg(synth.synth_frame(75, i, 640, 480)) for i in range(4)]
        m.profile_reset()
        _, st = m.infer_jpeg_batch(jpegs)
        assert st == [0] * 4
        names = {p["name"] for p in m.profile_read()}
        assert "huff_write" in names and "h2d_coef" not in names, names
    finally:
        m.close()


def test_default_handle_routes_entropy_by_stream_kind(weights, oracle_lib):
    """No flags: baseline single-scan streams, with or without restart markers, take the GPU
    entropy kernels, progressive streams the host workers; all decode to the same detections."""
    from infercam_onnx_amd import synth

    m = make_model(640, weights, max_batch=4, profile=True)
    try:
        frames = [synth.synth_frame(77, i, 640, 480) for i in range(3)]
        seen = []
        results = []
        for kw in ({}, {"restart_rows": 1}, {"progressive": True}):
            jpegs = [synth.encode_jpeg(f, **kw) for f in frames]
            m.profile_reset()
            res, st = m.infer_jpeg_batch(jpegs)
            assert st == [0] * 3
            seen.append({p["name"] for p in m.profile_read() if p["launches"] > 0})
            results.append(res)
        assert "huff_write" in seen[0] and "h2d_coef" not in seen[0]
        assert "huff_write" in seen[1] and "h2d_coef" not in seen[1] and "huffman_rst" not in seen[1]
        assert "h2d_coef" in seen[2] and "huff_write" not in seen[2]
        assert results[0] == results[1]  # same pixels (baseline, same quantisation), same kernels
    finally:
        m.close()


def test_device_sync_decoder_truncated_stream_is_flagged(model640_dev, oracle_lib):
    from infercam_onnx_amd import nn, synth

    good = synth.encode_jpeg(synth.synth_frame(76, 0, 640, 480))
    cut = good[: len(good) // 2] + b"\xff\xd9"  # half the scan data, then EOI
    res, status = model640_dev.infer_jpeg_batch([good, cut, good])
    assert status[0] == 0 and status[2] == 0 and res[0] == res[2]
    assert status[1] == nn.UFD_E_DECODE


def test_device_entropy_end_to_end_matches_host_entropy(model640, model640_dev, oracle_lib, weights):
    from infercam_onnx_amd import synth

    for kw in ({"restart_rows": 1}, {}):
        jpegs = [synth.encode_jpeg(synth.synth_frame(73, i, 640, 480), **kw) for i in range(6)]
        host, st_h = model640.infer_jpeg_batch(jpegs)
        dev, st_d = model640_dev.infer_jpeg_batch(jpegs)
        assert st_h == st_d == [0] * 6 and host == dev  # same batch size, same kernels: bit-identical
    prof = [p["name"] for p in model640_dev.profile_read()] if False else None  # (profiling is per-handle opt-in)


def test_device_entropy_corrupt_interval_is_flagged(model640_dev, oracle_lib, weights):
    model640 = model640_dev
    from infercam_onnx_amd import nn, synth

    good = synth.encode_jpeg(synth.synth_frame(71, 0, 640, 480), restart_rows=1)
    sos = good.index(b"\xff\xda")
    bad = bytearray(good)
    for i in range(sos + 200, sos + 260):  # garbage inside the first intervals (keep marker structure)
        if bad[i] != 0xFF and bad[i - 1] != 0xFF:
            bad[i] = 0x55
    res, status = model640.infer_jpeg_batch([good, bytes(bad), good])
    assert status[0] == 0 and status[2] == 0 and res[0] == res[2]
    # the reference would panic on any libjpeg warning; here the frame is either skipped or decoded
    assert status[1] in (0, nn.UFD_E_DECODE)
    pri = synth.gen_priors(640, 480)
    ref = oracle_lib.infer_jpeg(good, 640, 480, weights, pri, 0.5, 0.5)
    x = oracle_lib.normalize_nchw(oracle_lib.jpeg_decode_rgb(good))
    scores, _ = oracle_lib.forward(x, weights, pri)
    assert_dets_match(dets_array(res[0]), ref, scores=scores)


def test_mixed_batch_restart_and_plain_streams(model640_dev, oracle_lib):
    model640 = model640_dev
    from infercam_onnx_amd import synth

    f = synth.synth_frame(72, 0, 640, 480)
    a = synth.encode_jpeg(f, restart_rows=1)
    b = synth.encode_jpeg(f)  # no DRI
    res, status = model640.infer_jpeg_batch([a, b, a])
    assert status == [0, 0, 0] and res[0] == res[1] == res[2]


def test_staged_batch_matches_host_boundary(model640_dev, weights):
    """HBM-resident input (ufd_stage_jpeg_batch / ufd_submit_staged) gives the detections of the
    host-buffer boundary, and a staged batch can be submitted repeatedly."""
    from infercam_onnx_amd import nn, synth

    jpegs = [synth.encode_jpeg(synth.synth_frame(81, i, 640, 480)) for i in range(5)]
    jpegs[2] = jpegs[2][: len(jpegs[2]) // 3]  # no EOI: skipped at staging
    ref, st_ref = model640_dev.infer_jpeg_batch(jpegs)
    b = model640_dev.stage_jpeg_batch(jpegs)
    try:
        for _ in range(3):
            got, st = model640_dev.wait(model640_dev.submit_staged(b))
            assert st == st_ref and st[2] == nn.UFD_E_DECODE
            assert got == ref
    finally:
        model640_dev.free_staged(b)


def test_staged_batch_with_restart_markers(model640_dev):
    from infercam_onnx_amd import synth

    jpegs = [synth.encode_jpeg(synth.synth_frame(82, i, 640, 480), restart_rows=1) for i in range(3)]
    ref, st_ref = model640_dev.infer_jpeg_batch(jpegs)
    b = model640_dev.stage_jpeg_batch(jpegs)
    try:
        got, st = model640_dev.wait(model640_dev.submit_staged(b))
        assert st == st_ref == [0, 0, 0] and got == ref
    finally:
        model640_dev.free_staged(b)


def test_staging_rejects_ineligible_input(model640_dev, weights):
    from infercam_onnx_amd import nn, synth

    model640 = make_model(640, weights, max_batch=2, host_entropy=True)

    prog = synth.encode_jpeg(synth.synth_frame(83, 0, 640, 480), progressive=True)
    with pytest.raises(nn.UfdError) as e:
        model640_dev.stage_jpeg_batch([prog])
    assert e.value.code == nn.UFD_E_UNSUPPORTED
    try:
        with pytest.raises(nn.UfdError) as e:  # handle without the device entropy decoder
            model640.stage_jpeg_batch([synth.encode_jpeg(synth.synth_frame(83, 1, 640, 480))])
        assert e.value.code == nn.UFD_E_STATE
    finally:
        model640.close()


@pytest.mark.parametrize("variant", [320, 640])
def test_chained_blocks_kernel_is_bit_identical_to_the_unfused_pair(weights, oracle_lib, variant):
    """m1 -> m2 and m3 -> m4 run as one launch each (k_dwpw2_mfma, the 32-channel tensor between the
    blocks never exists) and keep the unfused fma order: scores and boxes are bit-identical to the two-launch path, for
    frames whose borders exercise the zero padding, and at batch sizes that leave dead lanes."""
    from infercam_onnx_amd import synth

    W, H = (640, 480) if variant == 640 else (320, 240)
    x = np.stack([oracle_lib.normalize_nchw(synth.synth_frame(91, i, W, H)) for i in range(3)])
    ref_model = make_model(variant, weights, max_batch=3, profile=True, no_chain=True)
    fused_model = make_model(variant, weights, max_batch=3, profile=True)
    try:
        for count in (3, 1):
            s0, b0 = ref_model.debug_forward(x[:count])
            s1, b1 = fused_model.debug_forward(x[:count])
            if count == 3 and variant == 640:
                # (smaller launches -- batch 1, or m4 of the 320 model -- are split-K in the unfused
                # form: fp32 rounding apart)
                assert np.array_equal(s0, s1) and np.array_equal(b0, b1)
            else:
                assert np.abs(s0 - s1).max() <= 5e-6 and np.abs(b0 - b1).max() <= 5e-6
        names_ref = {p["name"] for p in ref_model.profile_read() if p["launches"]}
        names_fused = {p["name"] for p in fused_model.profile_read() if p["launches"]}
        assert any(n.startswith("conv_dwpw2_mfma") for n in names_fused), names_fused
        assert not any(n.startswith("conv_dwpw2_mfma") for n in names_ref)
    finally:
        ref_model.close()
        fused_model.close()


@pytest.mark.parametrize("variant,batch", [(640, 32), (640, 5), (320, 32), (320, 16)])
def test_chained_blocks_kernel_row_rolling_at_bench_batches(weights, oracle_lib, variant, batch):
    """The chained kernel walks bands of output rows whose height the launcher picks from the launch size (row rolling:
    5 rows at batch 32 of the 640 model, other divisors of the map height elsewhere).  Whatever the band, m2.pw's and
    m4.pw's outputs must equal the two-launch path's bit for bit (same fma order), bands, frames and image borders
    straddled by one wave included."""
    from infercam_onnx_amd import synth

    W, H = (640, 480) if variant == 640 else (320, 240)
    base = np.stack([oracle_lib.normalize_nchw(synth.synth_frame(97, i, W, H)) for i in range(4)])
    x = np.concatenate([base] * ((batch + 3) // 4))[:batch].copy()
    x[1::2] = x[1::2][:, :, ::-1, :]  # (different content per frame: flipped copies)
    ref_model = make_model(variant, weights, max_batch=batch, tap_layers=True, no_chain=True)
    fused_model = make_model(variant, weights, max_batch=batch, tap_layers=True)
    try:
        ref_model.debug_forward(x)
        fused_model.debug_forward(x)
        for layer in (4, 8):  # m2.pw, m4.pw: the outputs of the two chained launches
            for frame in (0, 1, batch // 2, batch - 1):
                a, b = ref_model.debug_layer_output(layer, frame), fused_model.debug_layer_output(layer, frame)
                if variant == 640 or layer == 4:
                    assert np.array_equal(a, b), (layer, frame)
                else:  # (m4 of the 320 model is split-K in the unfused form: fp32 rounding apart)
                    assert np.abs(a - b).max() <= 5e-6, (layer, frame)
    finally:
        ref_model.close()
        fused_model.close()


@pytest.fixture(scope="module")
def model320_auto(weights):
    m = make_model(320, weights, max_batch=4)  # no entropy flags: the product default
    yield m
    m.close()


@pytest.mark.parametrize("size", [(640, 480), (641, 479), (1280, 720), (17, 9), (8, 8), (33, 100)])
@pytest.mark.parametrize("subsampling", ["4:2:0", "4:2:2", "4:4:4"])
def test_restart_interval_streams_through_the_segment_pipeline(model320_auto, oracle_lib, size, subsampling):
    """Restart intervals are segments with exact entry states in the self-synchronising decoder
    (DC predictors restart, padding bits before every marker): same pixels as the oracle for one
    or several MCU rows per interval, long and short symbols, noise."""
    from infercam_onnx_amd import synth

    w, h = size
    rgb = synth.synth_frame(15, w * 7 + h, w, h)
    noise = np.random.default_rng(w + h).integers(0, 256, size=rgb.shape, dtype=np.uint8)
    for img, kw in ((rgb, {"restart_rows": 1}), (rgb, {"restart_rows": 2}), (rgb, {"restart_rows": 1, "quality": 30}),
                    (rgb, {"restart_rows": 3, "optimize": True}), (noise, {"restart_rows": 1, "quality": 95}),
                    (np.zeros_like(rgb), {"restart_rows": 1})):
        jpeg = synth.encode_jpeg(img, subsampling=subsampling, **kw)
        assert b"\xff\xdd" in jpeg
        got = model320_auto.debug_decode_jpeg(jpeg)
        assert np.array_equal(got, oracle_lib.jpeg_decode_rgb(jpeg)), (size, subsampling, kw)


def test_mixed_stream_kinds_in_one_batch_on_the_device(weights, oracle_lib):
    from infercam_onnx_amd import synth

    m = make_model(640, weights, max_batch=4, profile=True)
    try:
        f = synth.synth_frame(78, 0, 640, 480)
        jpegs = [synth.encode_jpeg(f, restart_rows=1), synth.encode_jpeg(f), synth.encode_jpeg(f, restart_rows=4),
                 synth.encode_jpeg(f, subsampling="4:2:2")]
        res, st = m.infer_jpeg_batch(jpegs)
        names = {p["name"] for p in m.profile_read() if p["launches"] > 0}
        assert st == [0] * 4 and "huff_write" in names and "h2d_coef" not in names
        assert res[0] == res[1] == res[2]
        ref = [oracle_lib.jpeg_decode_rgb(j) for j in jpegs]
        assert np.array_equal(ref[0], ref[1]) and np.array_equal(ref[0], ref[2])
    finally:
        m.close()


def test_corrupt_restart_interval_is_flagged_by_the_segment_pipeline(model320_auto):
    from infercam_onnx_amd import nn, synth

    good = synth.encode_jpeg(synth.synth_frame(79, 0, 320, 240), restart_rows=1)
    sos = good.index(b"\xff\xda")
    bad = bytearray(good)
    for i in range(sos + 100, sos + 400):  # garbage inside the first intervals (markers stay in place)
        if bad[i] != 0xFF and bad[i - 1] != 0xFF and bad[i + 1] != 0xFF:
            bad[i] = 0x00
    res, status = model320_auto.infer_jpeg_batch([good, bytes(bad), good])
    assert status[0] == 0 and status[2] == 0 and res[0] == res[2]
    assert status[1] in (0, nn.UFD_E_DECODE)  # decoded (garbage pixels) or skipped, never a crash


def test_corrupted_entropy_segments_never_break_the_handle(model640_dev, oracle_lib):
    """Random damage inside the entropy-coded segment (byte flips, zero runs, 0xFF injections,
    truncation) across stream kinds: every frame ends as decoded or UFD_E_DECODE / UNSUPPORTED,
    nothing hangs or faults, and the same handle still decodes a clean batch bit-exactly."""
    from infercam_onnx_amd import nn, synth

    rng = np.random.default_rng(2024)
    clean = [synth.encode_jpeg(synth.synth_frame(90, i, 640, 480), **kw)
             for i, kw in enumerate(({}, {"restart_rows": 1}, {"subsampling": "4:2:2"}, {"restart_rows": 2, "quality": 40}))]
    ref, st_ref = model640_dev.infer_jpeg_batch(clean)
    assert st_ref == [0] * 4
    allowed = (0, nn.UFD_E_DECODE, nn.UFD_E_UNSUPPORTED)
    for rnd in range(12):
        batch = []
        for j in clean:
            b = bytearray(j)
            sos = j.index(b"\xff\xda") + 14
            kind = rng.integers(0, 5)
            pos = int(rng.integers(sos, len(b) - 2))
            if kind == 0:
                for p in rng.integers(sos, len(b) - 2, size=8):
                    b[int(p)] ^= int(rng.integers(1, 256))
            elif kind == 1:
                b[pos:pos + 64] = bytes(64)
            elif kind == 2:
                b[pos:pos + 3] = b"\xff\xff\xff"
            elif kind == 3:
                b = b[:pos] + b[-2:]  # truncated, EOI kept
            else:
                b[pos:pos + 2] = b"\xff\xd3"  # a stray restart marker
            batch.append(bytes(b))
        res, st = model640_dev.infer_jpeg_batch(batch)
        assert all(s in allowed for s in st), st
    again, st_again = model640_dev.infer_jpeg_batch(clean)
    assert st_again == [0] * 4 and again == ref


def test_stem_from_planes_is_bit_identical_to_the_two_kernel_path(weights, oracle_lib):
    """4:2:0 frames at the model size: the stem conv reads the decoder's sample planes itself
    (k_stem_planes_mfma: upsampling + colour + normalisation per lane).  Same integer formulas and
    MFMA order as k_upsample_norm_420 + the row kernel: detections are bit-identical, for frames
    whose first/last rows and columns exercise the padding, with a failed frame in the batch."""
    from infercam_onnx_amd import synth

    jpegs = [synth.encode_jpeg(synth.synth_frame(92, i, 640, 480), **kw)
             for i, kw in enumerate(({}, {"restart_rows": 1}, {"quality": 35}, {"quality": 98}))]
    jpegs.insert(2, jpegs[0][: len(jpegs[0]) // 2])  # a frame that fails to decode
    ref_model = make_model(640, weights, max_batch=5, profile=True, no_stem_fuse=True)
    ref, st_ref = ref_model.infer_jpeg_batch(jpegs)
    fused_model = make_model(640, weights, max_batch=5, profile=True)
    try:
        got, st = fused_model.infer_jpeg_batch(jpegs)
        assert st == st_ref and st[2] != 0 and got == ref
        names_ref = {p["name"] for p in ref_model.profile_read() if p["launches"]}
        names = {p["name"] for p in fused_model.profile_read() if p["launches"]}
        assert any(n.startswith("stem_planes_mfma:") for n in names) and "upsample_norm_420" not in names, names
        assert "upsample_norm_420" in names_ref and not any(n.startswith("stem_planes_mfma:") for n in names_ref)
    finally:
        ref_model.close()
        fused_model.close()


def test_stem_from_422_planes_is_bit_identical_to_the_two_kernel_path(weights, oracle_lib):
    """The twin for 4:2:2 (h2v1: chroma at full height, horizontal fancy upsampling only -- the UVC-MJPG flavour the
    reference's sender captures, cam_sender/src/sensors.rs:18-68), round 5: the fused stem picks the upsampling per FRAME
    from its descriptor, so a batch may mix 4:2:2 and 4:2:0 frames; without DHT segments, with restart markers, and with a
    failed frame in the batch.  Detections bit-identical to k_upsample_norm + the row kernel, and equal to the oracle's."""
    from infercam_onnx_amd import synth

    def enc(i, **kw):
        return synth.encode_jpeg(synth.synth_frame(93, i, 640, 480), **kw)

    jpegs = [enc(0, subsampling="4:2:2"), enc(1, subsampling="4:2:2", restart_rows=1), synth.strip_dht(enc(2, subsampling="4:2:2")),
             enc(3, subsampling="4:2:0"), enc(4, subsampling="4:2:2", quality=35), enc(5, subsampling="4:2:2", quality=98)]
    jpegs.insert(2, jpegs[0][: len(jpegs[0]) // 2])  # a frame that fails to decode
    ref_model = make_model(640, weights, max_batch=7, profile=True, no_stem_fuse=True)
    ref, st_ref = ref_model.infer_jpeg_batch(jpegs)
    fused_model = make_model(640, weights, max_batch=7, profile=True)
    try:
        got, st = fused_model.infer_jpeg_batch(jpegs)
        assert st == st_ref and st[2] != 0 and got == ref
        names_ref = {p["name"] for p in ref_model.profile_read() if p["launches"]}
        names = {p["name"] for p in fused_model.profile_read() if p["launches"]}
        assert any(n.startswith("stem_planes_mfma:") for n in names) and not any(n.startswith("upsample_norm") for n in names), names
        assert "upsample_norm" in names_ref and not any(n.startswith("stem_planes_mfma:") for n in names_ref)
        pri = synth.gen_priors(640, 480)
        for i, j in enumerate(jpegs):
            if st[i] == 0:
                assert_dets_match(dets_array(got[i]), oracle_lib.infer_jpeg(j, 640, 480, weights, pri, 0.5, 0.5), what="frame %d" % i)
        # all frames 4:2:2 (a camera stream has one flavour): same path
        only422 = [j for i, j in enumerate(jpegs) if i not in (2, 4)]
        got2, st2 = fused_model.infer_jpeg_batch(only422)
        assert st2 == [0] * 5
        for g2, g in zip(got2, [g for i, g in enumerate(got) if i not in (2, 4)]):  # (another batch size may pick other kernel
            assert_dets_match(dets_array(g2), dets_array(g), what="4:2:2-only batch")  # instances: fp32 rounding apart)
    finally:
        ref_model.close()
        fused_model.close()


@pytest.mark.parametrize("variant,batch", [(640, 4), (320, 17), (640, 32)])
def test_rfb_tail_launch_matches_the_two_launch_form(weights, oracle_lib, variant, batch):
    """k_rfb_tail (round 4): the three dilated 3x3 convs of the RFB branches hand their results to
    relu(ConvLinear(cat) + shortcut(x)) in registers -- the 48-channel concat tensor never exists, one launch instead of
    two.  Same MFMA sequence for the 3x3 convs; the 1x1 takes the branch channels in the accumulators' order, so fp32
    rounding apart (<= 5e-6 on scores / boxes) from the two-launch form (UFD_FLAG_NO_RFB_TAIL), the RFB output within the
    usual 1e-5 of the oracle on every frame, at tile counts that end inside a frame (320: 30x40 maps) and at the bench's
    batch.  (Handles for fewer than four 60x80 maps' worth of pixels keep the two launches: plan.cpp, tail_pays.)"""
    from infercam_onnx_amd import nn, synth

    W, H = (640, 480) if variant == 640 else (320, 240)
    pri = synth.gen_priors(W, H)
    x = np.stack([oracle_lib.normalize_nchw(synth.synth_frame(94, i % 7, W, H)) for i in range(batch)])
    ref_model = make_model(variant, weights, max_batch=batch, profile=True, extra_flags=nn.UFD_FLAG_NO_RFB_TAIL)
    fused_model = make_model(variant, weights, max_batch=batch, profile=True, tap_layers=True)
    try:
        s0, b0 = ref_model.debug_forward(x)
        s1, b1 = fused_model.debug_forward(x)
        assert np.abs(s0 - s1).max() <= 5e-6 and np.abs(b0 - b1).max() <= 5e-6
        for f in sorted({0, batch // 2, batch - 1}):
            rs, rb, outs = oracle_lib.forward(x[f], weights, pri, layers=True)
            assert np.abs(s1[f] - rs).max() <= 1e-5 and np.abs(b1[f] - rb).max() <= 1e-5
            got = fused_model.debug_layer_output(24, f).reshape(outs[24].shape)
            assert np.abs(got - outs[24]).max() <= 1e-5 * max(np.abs(outs[24]).max(), 1e-6)
            for gone in (15, 18, 22, 23):  # computed inside the launch: no tensor of their own in this plan
                with pytest.raises(nn.UfdError):
                    fused_model.debug_layer_output(gone, f)
        names_ref = {p["name"] for p in ref_model.profile_read() if p["launches"]}
        names = {p["name"] for p in fused_model.profile_read() if p["launches"]}
        assert any(n.startswith("rfb_tail:") for n in names) and not any(n.startswith("rfb_tail:") for n in names_ref), names
        assert any(n.startswith("conv3x3_rows_mfma<1, 0>") for n in names_ref) and not any(n.startswith("conv3x3_rows_mfma<1, 0>") for n in names)
    finally:
        ref_model.close()
        fused_model.close()


def test_summed_rfb_convs_match_the_two_launch_form(weights, oracle_lib):
    """relu(ConvLinear(cat) + shortcut(x)) as one 1x1 conv over both inputs' channels: one fma chain
    instead of two, so fp32 rounding apart (<= 5e-6 on scores / boxes) from the two-launch form,
    and within the usual bar of the oracle."""
    from infercam_onnx_amd import synth

    W, H = 640, 480
    pri = synth.gen_priors(W, H)
    x = np.stack([oracle_lib.normalize_nchw(synth.synth_frame(93, i, W, H)) for i in range(3)])
    from infercam_onnx_amd import nn

    ref_model = make_model(640, weights, max_batch=3, profile=True, no_rfb_sum=True)
    fused_model = make_model(640, weights, max_batch=3, profile=True, extra_flags=nn.UFD_FLAG_NO_RFB_TAIL)  # (the summed conv as its own launch)
    try:
        s0, b0 = ref_model.debug_forward(x)
        s1, b1 = fused_model.debug_forward(x)
        assert np.abs(s0 - s1).max() <= 5e-6 and np.abs(b0 - b1).max() <= 5e-6
        rs, rb = oracle_lib.forward(x[0], weights, pri)
        assert np.abs(s1[0] - rs).max() <= 1e-5 and np.abs(b1[0] - rb).max() <= 1e-5
        names_ref = {p["name"] for p in ref_model.profile_read() if p["launches"]}
        names = {p["name"] for p in fused_model.profile_read() if p["launches"]}
        assert any("rfb.linear" in n for n in names_ref) and not any("rfb.linear" in n for n in names), names
    finally:
        ref_model.close()
        fused_model.close()


@pytest.mark.parametrize("variant,batch", [(640, 32), (640, 3), (320, 32), (320, 1)])
def test_dual_launches_are_bit_identical_to_separate_ones(weights, oracle_lib, variant, batch):
    """A cls/reg head pair and the backbone block beside it as ONE grid (k_dual_*): same bodies, same numbering inside
    each conv, so every tensor of the issued plan equals the separate-launch plan bit for bit -- at the bench's batch
    (plain instances) and at small batches (split-K instances, or pairs that are not compiled and fall back)."""
    from infercam_onnx_amd import synth

    W, H = VARIANT_WH[variant]
    x = np.stack([oracle_lib.normalize_nchw(synth.synth_frame(97, i, W, H)) for i in range(min(batch, 3))])
    x = np.concatenate([x] * ((batch + len(x) - 1) // len(x)))[:batch]
    ref_model = make_model(variant, weights, max_batch=batch, profile=True, tap_layers=True, no_dual=True)
    dual_model = make_model(variant, weights, max_batch=batch, profile=True, tap_layers=True)
    try:
        s0, b0 = ref_model.debug_forward(x)
        s1, b1 = dual_model.debug_forward(x)
        assert np.array_equal(s0, s1) and np.array_equal(b0, b1)
        for layer in (26, 28, 30, 36, 38, 40, 44, 46, 47):
            for f in (0, batch - 1):
                assert np.array_equal(ref_model.debug_layer_output(layer, f), dual_model.debug_layer_output(layer, f)), (layer, f)
        names_ref = {p["name"] for p in ref_model.profile_read() if p["launches"]}
        names = {p["name"] for p in dual_model.profile_read() if p["launches"]}
        assert not any(n.startswith("conv_dual") for n in names_ref)
        if (variant, batch) == (640, 32):  # the bench's plan: all three pairs ride
            assert sum(n.startswith("conv_dual") for n in names) == 3, names
    finally:
        ref_model.close()
        dual_model.close()


def test_unusual_restart_layouts(model320_auto, oracle_lib):
    """Restart intervals of a single MCU (1200 per frame: more than the device pipeline's segment
    table, so the batch decodes on the host workers), of a few MCUs (segments shorter than one
    subsequence slot) and grayscale streams with restart markers: same pixels as the oracle."""
    import io
    from PIL import Image
    from infercam_onnx_amd import synth

    rgb = synth.synth_frame(16, 5, 640, 480)
    for mode, kw in (("RGB", {"restart_marker_blocks": 1}), ("RGB", {"restart_marker_blocks": 3}),
                     ("RGB", {"restart_marker_blocks": 7, "subsampling": 0}), ("L", {"restart_marker_rows": 1}),
                     ("L", {"restart_marker_blocks": 5})):
        buf = io.BytesIO()
        Image.fromarray(rgb).convert(mode).save(buf, format="JPEG", quality=88, **kw)
        jpeg = buf.getvalue()
        assert b"\xff\xdd" in jpeg
        got = model320_auto.debug_decode_jpeg(jpeg)
        assert np.array_equal(got, oracle_lib.jpeg_decode_rgb(jpeg)), (mode, kw)


def test_table_set_and_tap_caches_evict(weights, oracle_lib):
    """More distinct Huffman table sets (per-frame optimised tables) than the handle caches (64), and more distinct
    frame sizes than it keeps resize taps for (32): old entries are evicted, every frame still decodes bit-exactly on
    the device path and detections still match the oracle."""
    from infercam_onnx_amd import synth

    model = make_model(320, weights, max_batch=4, profile=True)
    try:
        rng = np.random.default_rng(8)
        early = []
        for k in range(80):  # optimize=True: Huffman tables fitted to each frame
            w, h = 64 + 8 * (k % 40), 48 + 8 * ((k * 7) % 23)
            rgb = synth.synth_frame(500 + k, k, w, h)
            rgb = np.clip(rgb.astype(np.int16) + rng.integers(-40, 40, rgb.shape), 0, 255).astype(np.uint8)
            j = synth.encode_jpeg(rgb, quality=60 + k % 35, optimize=True)
            assert np.array_equal(model.debug_decode_jpeg(j), oracle_lib.jpeg_decode_rgb(j)), k
            if k < 12:
                early.append(j)
            if k % 8 == 0:
                got = dets_array(model.infer_jpeg(j))
                assert_dets_match(got, oracle_lib.infer_jpeg(j, 320, 240, weights, synth.gen_priors(320, 240)), what="evict %d" % k)
        # streams whose table set has been evicted meanwhile come back (round 4: sets are found by the key of their DHT bytes;
        # a key whose set was replaced must miss, the tables be rebuilt and the key point at the new slot), twice over
        for rep in range(2):
            for k, j in enumerate(early):
                assert np.array_equal(model.debug_decode_jpeg(j), oracle_lib.jpeg_decode_rgb(j)), ("back", rep, k)
        names = {p["name"] for p in model.profile_read() if p["launches"]}
        assert any(n.startswith("huff_write") for n in names), names  # the device entropy path stayed in use
    finally:
        model.close()


# ---------------------------------------------------------------- A1: every layout libjpeg-turbo decodes (round 6)
def _layout_fixtures():
    from test_oracle_jpeg import layout_fixtures

    return layout_fixtures()


def _check_layout_stream(m, oracle_lib, name, kind, jpeg, rgb, sha):
    from test_oracle_jpeg import check_pixels

    got = m.debug_decode_jpeg(jpeg)
    ref = oracle_lib.jpeg_decode_rgb(jpeg)
    assert got.shape == ref.shape and np.array_equal(got, ref), "%s/%s: %d samples differ from the oracle" % (name, kind, (got != ref).sum())
    check_pixels(name, kind, got, rgb, sha)  # ... and from what libjpeg-turbo itself decoded


@pytest.mark.parametrize("path", ["host_entropy", "device_entropy", "default"])
def test_every_libjpeg_layout_on_the_gpu(model320, model320_dev, model320_auto, oracle_lib, path):
    """`turbojpeg::decompress_image` (inferer.rs:35) is libjpeg-turbo: it takes any integral sampling layout.  The 329
    libjpeg-turbo-written streams of tests/golden/jpeg_layouts.npz (4:4:0, 4:1:1, 4:1:0, 4:4:1, 3x1 ..., luma coarser
    than chroma, a rate per chroma plane, RGB colour space by marker rules, grey with a 2x2 SOF; baseline / two restart
    layouts / progressive / optimised tables) decode on the GPU to the oracle's and to libjpeg-turbo's pixels, with the
    entropy stage on the host workers, on the device, and wherever the product's default routing sends it; round 6 added the
    NON-INTERLEAVED files (a scan per component, luma + chroma pair with restart intervals: the host workers' multi-scan path)."""
    m = {"host_entropy": model320, "device_entropy": model320_dev, "default": model320_auto}[path]
    n = 0
    for name, (streams, rgb, sha) in sorted(_layout_fixtures().items()):
        for kind, jpeg in streams.items():
            _check_layout_stream(m, oracle_lib, name, kind, jpeg, rgb, sha)
            n += 1
    from test_oracle_jpeg import N_LAYOUT_STREAMS

    assert n == N_LAYOUT_STREAMS


def test_layouts_take_the_device_entropy_decoder(weights, oracle_lib):
    """Baseline 4:1:1 / 4:1:0 (10 blocks per MCU) / 4:4:0 / RGB-colourspace frames are decoded by the self-synchronising
    DEVICE decoder, not by the host fallback (the profile shows huff_write and no coefficient upload), mixed in one batch
    with a 4:2:0 frame; detections equal the oracle's."""
    from infercam_onnx_amd import synth

    fx = _layout_fixtures()
    m = make_model(640, weights, max_batch=8, profile=True)
    try:
        jpegs = [fx["411_640x480"][0]["base"], fx["410_640x480"][0]["dri_row"], fx["440_640x480"][0]["base"],
                 synth.encode_jpeg(synth.synth_frame(21, 1120, 640, 480)), fx["rgb_adobe0_320x240"][0]["base"],
                 fx["rgb_420_320x240"][0]["dri_row"], fx["y11c22_150x100"][0]["base"], fx["gray_sof22_67x45"][0]["base"]]
        res, st = m.infer_jpeg_batch(jpegs)
        names = {p["name"] for p in m.profile_read() if p["launches"] > 0}
        assert st == [0] * len(jpegs) and "huff_write" in names and "h2d_coef" not in names, (st, sorted(names))
        pri = synth.gen_priors(640, 480)
        for j, r in zip(jpegs, res):
            x = oracle_lib.normalize_nchw(oracle_lib.resize_triangle(oracle_lib.jpeg_decode_rgb(j), 640, 480))
            scores, _ = oracle_lib.forward(x, weights, pri)
            assert_dets_match(dets_array(r), oracle_lib.infer_jpeg(j, 640, 480, weights, pri, 0.5, 0.5), scores=scores)
    finally:
        m.close()


def test_layouts_libjpeg_refuses_are_decode_errors_on_the_gpu_path(model320_auto):
    """Fractional expansion and more than 10 blocks per MCU: libjpeg-turbo fails, the reference panics (inferer.rs:35-36);
    the product skips the frame with a status and the frames beside it are untouched."""
    from infercam_onnx_amd import nn, synth
    from test_oracle_jpeg import _patch_sof

    good = synth.encode_jpeg(synth.synth_frame(3, 0, 64, 48), subsampling="4:4:4")
    bad = [_patch_sof(_patch_sof(good, 0, 0x31), 1, 0x21), _patch_sof(_patch_sof(good, 0, 0x42), 1, 0x21)]
    res, status = model320_auto.infer_jpeg_batch([good, bad[0], bad[1], good])
    assert status[0] == 0 and status[3] == 0 and res[0] == res[3]
    assert status[1] in (nn.UFD_E_UNSUPPORTED, nn.UFD_E_DECODE) and status[2] == nn.UFD_E_DECODE and res[1] is None and res[2] is None


# ---------------------------------------------------------------- round 6: one launch for A7-A10, the gate, launch shapes
def test_small_batch_single_launch_nms_with_many_candidates(weights, oracle_lib):
    """A frame or a few at a time go in and out by kernels, and since round 6 A7-A10 plus the results' way out are ONE launch
    (k_sort_nms finishes every frame itself and writes statuses, counts and detections to the slot's pinned arrays).  With a
    low confidence threshold a frame has thousands of candidates -- the in-kernel block loop instead of the matrix path's
    two extra launches: same detections as the oracle, one frame and three at a time, and the profile shows neither
    k_nms_matrix, k_nms_scan nor a result copy."""
    from infercam_onnx_amd import nn, synth

    pri = synth.gen_priors(320, 240)
    jpegs = [synth.encode_jpeg(synth.synth_frame(61, i, 320, 240)) for i in range(3)]
    for min_conf, lo in ((0.5, 0), (0.2, 257), (0.03, 1000)):
        m = nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, min_conf, weights=weights, priors=pri, max_batch=4, max_src=(320, 240),
                              det_cap=4420, profile=True)
        try:
            refs = [oracle_lib.infer_jpeg(j, 320, 240, weights, pri, min_conf, 0.5) for j in jpegs]
            x = [oracle_lib.normalize_nchw(oracle_lib.jpeg_decode_rgb(j)) for j in jpegs]
            ncand = [int((oracle_lib.forward(xi, weights, pri)[0][..., 1] > min_conf).sum()) for xi in x]
            assert max(ncand) >= lo, ncand
            for j, r in zip(jpegs, refs):
                assert_dets_match(dets_array(m.infer_jpeg(j)), r, min_conf=min_conf, what="one frame, min_conf %g" % min_conf)
            t = m.submit_jpeg_batch(jpegs)
            got, st = m.wait(t)
            assert st == [0, 0, 0]
            for g, r in zip(got, refs):
                assert_dets_match(dets_array(g), r, min_conf=min_conf, what="three frames, min_conf %g" % min_conf)
            names = {p["name"].split(":")[0] for p in m.profile_read() if p["launches"] > 0}
            assert "sort_nms" in names and not names & {"nms_matrix", "nms_scan"}, sorted(names)
        finally:
            m.close()


def test_gate_orders_batches_without_changing_a_result(weights, oracle_lib):
    """csrc/pipeline_gate.cpp: batch n + 1's network waits for batch n's m3->m4 on another stream.  Twelve batches in flight
    six at a time, with and without UFD_FLAG_NO_GATE: identical detections (bit for bit: the same kernels on the same data),
    a batch with nothing decodable in the middle of the chain holds nobody up."""
    from infercam_onnx_amd import nn, synth

    pri = synth.gen_priors(640, 480)
    pool = [synth.encode_jpeg(synth.synth_frame(71, i, 640, 480)) for i in range(16)]
    batches = [pool[(4 * i) % 16:(4 * i) % 16 + 8] if i != 5 else [b"junk"] * 8 for i in range(12)]
    out = {}
    for flags in (0, nn.UFD_FLAG_NO_GATE):
        with nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=weights, priors=pri, max_batch=8, max_src=(640, 480),
                               det_cap=1024, extra_flags=flags) as m:
            res, infl = [], []
            for b in batches:
                if len(infl) >= 6:
                    res.append(m.wait(infl.pop(0)))
                infl.append(m.submit_jpeg_batch(b))
            res += [m.wait(t) for t in infl]
            out[flags] = res
    assert len(out[0]) == 12
    for (ra, sa), (rb, sb) in zip(out[0], out[nn.UFD_FLAG_NO_GATE]):
        assert sa == sb and ra == rb
    assert out[0][5][1] == [nn.UFD_E_DECODE] * 8 and out[0][6][1] == [0] * 8
    ref = oracle_lib.infer_jpeg(pool[0], 640, 480, weights, pri, 0.5, 0.5)
    assert_dets_match(dets_array(out[0][0][0][0]), ref)


def test_profile_shapes_report_how_launches_sit_on_the_gpu(weights):
    """ufd_profile_shapes: per profiled label the launch as issued (workgroups, threads), the kernel's registers and LDS and
    the workgroups a CU holds at once by the runtime's occupancy query -- DESIGN's kernel table is generated from it."""
    from infercam_onnx_amd import nn, synth

    with nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=weights, priors=synth.gen_priors(640, 480), max_batch=32,
                           max_src=(640, 480), det_cap=256, profile=True) as m:
        jpegs = [synth.encode_jpeg(synth.synth_frame(81, i, 640, 480)) for i in range(32)]
        m.wait(m.submit_jpeg_batch(jpegs))
        shapes = {s["name"]: s for s in m.profile_shapes()}
    assert len(shapes) >= 25
    for name, s in shapes.items():
        assert s["workgroups"] >= 1 and s["threads"] in (64, 128, 256, 512, 1024) and 1 <= s["resident_per_cu"] <= 32, (name, s)
        assert s["slots"] == 256 * s["resident_per_cu"] and s["registers"] > 0, (name, s)
    chained = [s for n, s in shapes.items() if n.startswith("conv_dwpw2_mfma<16")]
    assert chained and chained[0]["resident_per_cu"] == 2 and chained[0]["lds_bytes"] > 64 * 1024  # 80 KB of LDS, 256 registers: a CU whole
    coop = [s for n, s in shapes.items() if n.startswith("conv_dwpw_coop<1, 4>")]
    assert sorted(s["workgroups"] for s in coop) == [160, 320, 320]  # m12, m9, m10 at batch 32 (1.25 and 0.63 per CU)
