"""Oracle pinning, rows A2-A4: Triangle resize (image 0.24.5 sample.rs semantics) and
the normalisation closure (nn.rs:82-93), against analytic tap tables and an independent numpy twin."""
import numpy as np
import pytest


def axis_weights(S, D):
    """independent float32 numpy restatement of the per-axis window of sample.rs"""
    f = np.float32
    ratio = f(S) / f(D)
    sratio = max(ratio, f(1))
    out = []
    for o in range(D):
        c = (f(o) + f(0.5)) * ratio
        left = int(min(max(np.floor(c - sratio), 0), S - 1))
        right = int(min(max(np.ceil(c + sratio), left + 1), S))
        c = c - f(0.5)
        w = np.array([max(f(0), f(1) - abs((f(i) - c) / sratio)) for i in range(left, right)], np.float32)
        s = f(0)
        for x in w:
            s = f(s + x)
        out.append((left, (w / s).astype(np.float32)))
    return out


def numpy_resize(src, dw, dh):
    sh, sw, _ = src.shape
    if (sw, sh) == (dw, dh):
        return src.copy()
    tmp = np.zeros((dh, sw, 3), np.float32)
    for oy, (left, w) in enumerate(axis_weights(sh, dh)):
        acc = np.zeros((sw, 3), np.float32)
        for i, wi in enumerate(w):
            acc = (acc + src[left + i].astype(np.float32) * wi).astype(np.float32)
        tmp[oy] = acc
    out = np.zeros((dh, dw, 3), np.uint8)
    for ox, (left, w) in enumerate(axis_weights(sw, dw)):
        acc = np.zeros((dh, 3), np.float32)
        for i, wi in enumerate(w):
            acc = (acc + tmp[:, left + i] * wi).astype(np.float32)
        v = np.clip(acc, 0, 255)
        out[:, ox] = np.where(v - np.floor(v) >= 0.5, np.floor(v) + 1, np.floor(v)).astype(np.uint8)  # round half away
    return out


def test_tap_tables_from_survey():
    w = axis_weights(1280, 640)
    assert np.allclose(w[5][1], [0.125, 0.375, 0.375, 0.125]) and len(w[0][1]) == 3
    w = axis_weights(1280, 320)
    assert np.allclose(w[7][1], np.array([1, 3, 5, 7, 7, 5, 3, 1]) / 32)
    w = axis_weights(720, 240)
    assert np.allclose(w[9][1], np.array([0, 1, 2, 3, 2, 1, 0]) / 9, atol=1e-7)
    w = axis_weights(720, 480)
    assert np.allclose(sorted(w[10][1]), sorted([0, 1 / 3, 5 / 9, 1 / 9]), atol=1e-6)
    for left, ww in axis_weights(480, 480):
        assert np.count_nonzero(ww) == 1 and ww.max() == 1.0  # identity


@pytest.mark.parametrize("src,dst", [((128, 72), (64, 48)), ((64, 43), (64, 48)), ((37, 50), (64, 48)),
                                     ((256, 144), (32, 24)), ((64, 48), (64, 48))])
def test_resize_matches_numpy_twin(oracle_lib, src, dst):
    from infercam_onnx_amd import synth

    rgb = synth.synth_frame(9, src[0], src[0], src[1])
    got = oracle_lib.resize_triangle(rgb, dst[0], dst[1])
    assert np.array_equal(got, numpy_resize(rgb, dst[0], dst[1]))


def test_constant_image_and_identity(oracle_lib):
    rgb = np.full((90, 160, 3), 137, np.uint8)
    assert (oracle_lib.resize_triangle(rgb, 64, 48) == 137).all()
    from infercam_onnx_amd import synth

    f = synth.synth_frame(1, 2, 64, 48)
    assert np.array_equal(oracle_lib.resize_triangle(f, 64, 48), f)


@pytest.mark.parametrize("src,dst", [((1280, 720), (320, 240)), ((1280, 720), (640, 480)), ((640, 427), (640, 480)),
                                     ((100, 75), (320, 240)), ((333, 517), (320, 240))])
def test_resize_against_pillow_bilinear(oracle_lib, src, dst):
    """A resampler nobody here wrote: Pillow's BILINEAR is the same triangle window (support widened by the downscale
    ratio, centres at (o + 0.5) * ratio, weights normalised per output sample) with a different pass order and an
    8-bit intermediate, so the two agree to one grey level on every sample -- a wrong support, centre or clamp of the
    restated window would not."""
    from PIL import Image

    rng = np.random.default_rng(5)
    sw, sh = src
    y, x = np.mgrid[0:sh, 0:sw]
    img = np.stack([(128 + 100 * np.sin(x / 17.0 + c) * np.cos(y / 23.0) + rng.integers(-20, 20, (sh, sw))).clip(0, 255)
                    for c in range(3)], -1).astype(np.uint8)
    got = oracle_lib.resize_triangle(img, dst[0], dst[1])
    ref = np.asarray(Image.fromarray(img).resize(dst, Image.BILINEAR))
    d = np.abs(got.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 0.3


@pytest.mark.parametrize("src,dst", [((1280, 720), (640, 480)), ((640, 480), (320, 240)), ((800, 600), (640, 480)), ((322, 242), (320, 240)),
                                     ((100, 80), (320, 240)), ((1920, 1080), (640, 480)), ((641, 479), (640, 480))])
def test_resize_against_torch_antialiased_bilinear(oracle_lib, src, dst):
    """A second resampler nobody here wrote, and a much tighter one than Pillow's: ATen's antialiased bilinear
    (`interpolate(mode="bilinear", antialias=True, align_corners=False)`) is the same triangle window evaluated in
    floating point with NO 8-bit intermediate -- like image 0.24's f32 vertical pass.  Rounded the crate's way (clamp,
    round half away from zero) it equals the oracle on more than 99 % of all samples of every size pair and is never more
    than one grey level off (what differs are values that sit on x.5 exactly, where the order of the float additions
    decides: most often at the 2:1 ratio of 1280 -> 640).  Pillow leaves 30 % of the samples one level away."""
    import torch
    from infercam_onnx_amd import synth

    torch.set_num_threads(4)
    img = synth.synth_frame(1, 3, src[0], src[1])
    got = oracle_lib.resize_triangle(img, dst[0], dst[1]).astype(np.int32)
    t = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    r = torch.nn.functional.interpolate(t, size=(dst[1], dst[0]), mode="bilinear", antialias=True, align_corners=False)[0].permute(1, 2, 0)
    ref = torch.floor(r.clamp(0, 255) + 0.5).to(torch.int32).numpy()
    d = np.abs(got - ref)
    assert d.max() <= 1 and (d > 0).mean() < 0.01, (d.max(), (d > 0).mean())


def test_normalize_closure(oracle_lib):
    rgb = np.arange(256 * 3, dtype=np.uint32).reshape(16, 16, 3).astype(np.uint8)
    got = oracle_lib.normalize_nchw(rgb)
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    ref = ((rgb.astype(np.float32) / np.float32(255.0) - mean) / std).transpose(2, 0, 1)
    assert got.shape == (3, 16, 16) and np.array_equal(got, ref.astype(np.float32))


# ---------------------------------------------------------------------------------------------
# Exact-rational derivation of the Triangle windows, independent of the float32 numpy
# twin above: the window formula evaluated with Python fractions, then rounded ONCE to float32.
def exact_axis_window(S, D, o):
    from fractions import Fraction as F

    ratio = F(S, D)
    sratio = max(ratio, F(1))
    c = (F(o) + F(1, 2)) * ratio
    import math

    left = min(max(math.floor(c - sratio), 0), S - 1)
    right = min(max(math.ceil(c + sratio), left + 1), S)
    c = c - F(1, 2)
    w = [max(F(0), F(1) - abs((F(i) - c) / sratio)) for i in range(left, right)]
    tot = sum(w)
    return left, [x / tot for x in w]


# every axis of the seven source sizes x two model sizes of tests/test_gpu_parity.py::test_preproc_bit_exact
AXES = sorted({(s, d) for s in (1280, 640, 320, 100, 333) for d in (320, 640)} |
              {(s, d) for s in (720, 427, 960, 240, 480, 37, 500) for d in (240, 480)})


@pytest.mark.parametrize("S,D", AXES)
def test_oracle_windows_are_the_rounded_exact_rational_ones(oracle_lib, S, D):
    """The oracle's f32 weights (image 0.24.5 computes them in f32 too) against the exact rational weights, as a
    function over the source axis.  The f32 evaluation carries the rounding of `ratio` and of the centre
    c = (o + 0.5) * ratio, i.e. an absolute error of about ulp(c) <= S * 2^-23 in (i - c), divided by sratio: the
    bound asserted is 2 * 2^-23 * (2 + S / sratio) absolute (37 -> 240: 9e-6, i.e. 0.002 grey levels).  The weights
    sum to 1 within 2^-21, and where every quantity is dyadic (ratios 2 and 4) they are the exact values bit for bit."""
    dyadic = (S % D == 0) and ((S // D) & ((S // D) - 1)) == 0 and S != D
    worst = 0.0
    for o in range(D):
        left, w = oracle_lib.axis_taps(S, D, o)
        eleft, ew = exact_axis_window(S, D, o)
        dense = np.zeros(S, np.float64)
        dense[left:left + len(w)] = w
        edense = np.zeros(S, np.float64)
        edense[eleft:eleft + len(ew)] = [float(x) for x in ew]
        if S == D:  # image's resize() copies same-size images; the window degenerates to the pixel itself
            assert np.count_nonzero(dense) == 1 and dense[o] == 1.0
            continue
        worst = max(worst, np.abs(dense - edense).max())
        assert abs(float(np.sum(w, dtype=np.float64)) - 1.0) <= 2.0 ** -21
        if dyadic:
            assert left == eleft or dense[min(left, eleft):max(left, eleft)].sum() == 0
            assert np.array_equal(dense.astype(np.float32), edense.astype(np.float32)), (S, D, o)
    tol = 2.0 * 2.0 ** -23 * (2.0 + S / max(S / D, 1.0))
    assert worst <= tol, "S=%d D=%d: %.3g from the exact rational window (bound %.3g)" % (S, D, worst, tol)
