"""Row N1, CPU side: the oracle's rectangle drawing and JPEG encoder against libjpeg-turbo's own
streams (committed golden vectors + live encodes) and known-answer rectangles (inferer.rs:58-92, :39; lib.rs:48-57)."""
import importlib.util
import io
import os

import numpy as np
import pytest

import oracle
from infercam_onnx_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = np.load(os.path.join(ROOT, "tests", "golden", "encode_q95_420.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLDEN.files})


@pytest.mark.parametrize("name", NAMES)
def test_encoder_matches_libjpeg_turbo_streams(name):
    """tests/golden/encode_q95_420.npz: streams written by libjpeg-turbo 2.1.2 (tools/make_encode_golden.py) with
    tjCompress2's settings at quality 95 / 4:2:0 -- the fast DCT the reference gets, and the accurate one."""
    rgb = GOLDEN[name + "/rgb"]
    assert oracle.jpeg_encode_rgb(rgb, 95, 1) == GOLDEN[name + "/ifast"].tobytes()
    assert oracle.jpeg_encode_rgb(rgb, 95, 0) == GOLDEN[name + "/islow"].tobytes()
    assert oracle.jpeg_encode_rgb(rgb, 95, -1) == GOLDEN[name + "/ifast"].tobytes()  # below 96: JDCT_FASTEST


def _pil_islow(rgb, quality):
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", quality=quality, subsampling="4:2:0", optimize=False)
    return b.getvalue()


@pytest.mark.parametrize("size", [(640, 480), (320, 240), (333, 217), (17, 9), (8, 8), (136, 8)])
@pytest.mark.parametrize("quality", [95, 96, 75, 100])
def test_encoder_islow_equals_pil_live(size, quality):
    w, h = size
    rgb = synth.synth_frame(7, w + 3 * h, w, h)
    assert oracle.jpeg_encode_rgb(rgb, quality, 0) == _pil_islow(rgb, quality)
    if quality >= 96:  # tjCompress2 switches to the accurate DCT from 96 on
        assert oracle.jpeg_encode_rgb(rgb, quality, -1) == _pil_islow(rgb, quality)


def _turbo():
    spec = importlib.util.spec_from_file_location("make_encode_golden", os.path.join(ROOT, "tools", "make_encode_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.LIBJPEG):
        pytest.skip("no system libjpeg-turbo on this box")
    return mod, mod.TurboEncoder()


def test_encoder_ifast_equals_system_libjpeg_turbo_live():
    mod, enc = _turbo()
    rng = np.random.default_rng(5)
    frames = [synth.synth_frame(11, 3, 640, 480), synth.synth_frame(11, 4, 1280, 720)[:300, :500]]
    for _ in range(6):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
        frames.append(synth.synth_frame(3, w * h, max(w, 16), max(h, 16))[:h, :w])
    frames.append(rng.integers(0, 256, (64, 80, 3), dtype=np.uint8))  # saturated noise: the 16-bit wrap-around paths
    for rgb in frames:
        rgb = np.ascontiguousarray(rgb)
        assert oracle.jpeg_encode_rgb(rgb, 95, 1) == enc.encode(rgb, 95, mod.JDCT_IFAST), rgb.shape
        assert oracle.jpeg_encode_rgb(rgb, 50, 1) == enc.encode(rgb, 50, mod.JDCT_IFAST), rgb.shape


def test_encoded_stream_decodes_back():
    rgb = synth.synth_frame(2, 9, 320, 240)
    back = oracle.jpeg_decode_rgb(oracle.jpeg_encode_rgb(rgb, 95))
    err = np.abs(back.astype(np.int32) - rgb.astype(np.int32))
    assert back.shape == rgb.shape and err.mean() < 2.5 and err.max() < 48


def test_quant_tables_q95():
    # jpeg_set_quality(95): scale factor 10 -> (base * 10 + 50) / 100, clamped to [1, 255]
    assert list(oracle.quant_table(95, 0)[:8]) == [2, 1, 1, 2, 2, 4, 5, 6]
    assert list(oracle.quant_table(95, 1)[:8]) == [2, 2, 2, 5, 10, 10, 10, 10]


# ---- rectangles (inferer.rs:58-92) ----
GREEN = (0, 255, 0)


def _outline(shape, l, t, r, b):
    h, w = shape[:2]
    m = np.zeros((h, w), bool)
    for y in range(max(t, 0), min(b, h - 1) + 1):
        for x in range(max(l, 0), min(r, w - 1) + 1):
            if x in (l, r) or y in (t, b):
                m[y, x] = True
    return m


def test_rect_of_det_follows_the_reference_casts():
    # x_tl = 0.25 * 1280 = 320, y_tl = 0.5 * 720 = 360, width = (0.5 - 0.25) * 1280 = 320 -> right = 639; height 72 -> bottom 431
    assert oracle.rect_of_det([0.25, 0.5, 0.5, 0.6, 0.9], 1280, 720) == (320, 360, 639, 431)
    # fractions truncate toward zero (as i32 / as u32): x_tl = 10.9 -> 10, width 5.99 -> 5
    assert oracle.rect_of_det([10.9 / 100, 0.1, 16.89 / 100, 0.2, 0.9], 100, 100) == (10, 10, 14, 19)
    # negative corner truncates toward zero, not down: -0.7 -> 0
    assert oracle.rect_of_det([-0.007, -0.5, 0.5, 0.5, 0.9], 100, 100)[:2] == (0, -50)
    # width below one pixel (or negative): Rect::of_size would assert -> no rectangle
    assert oracle.rect_of_det([0.5, 0.5, 0.505, 0.9, 0.9], 100, 100) is None
    assert oracle.rect_of_det([0.6, 0.5, 0.5, 0.9, 0.9], 100, 100) is None
    assert oracle.rect_of_det([float("nan"), 0.5, 0.7, 0.9, 0.9], 100, 100) is None


@pytest.mark.parametrize("det,label", [
    ([0.25, 0.25, 0.75, 0.75, 0.9], (64, 48)),       # inside
    ([-0.2, 0.1, 0.5, 1.4, 0.8], (64, 48)),          # clipped left and bottom
    ([0.0, 0.0, 1.0, 1.0, 0.8], (64, 48)),           # the whole frame: right = 63, bottom = 47
    ([0.1, 0.1, 0.2, 0.2, 0.8], (1280, 720)),        # label size != frame size (router.rs:66-67): mostly outside
    ([0.5, 0.5, 0.5 + 1 / 64, 0.6, 0.8], (64, 48)),  # one pixel wide
])
def test_hollow_rect_pixels(det, label):
    rgb = synth.synth_frame(1, 1, 64, 48)
    out = oracle.draw_hollow_rects(rgb, [det], *label)
    rect = oracle.rect_of_det(det, *label)
    m = _outline(rgb.shape, *rect)
    assert (out[m] == GREEN).all()
    assert np.array_equal(out[~m], rgb[~m])


def test_several_rects_and_stream_item():
    rgb = synth.synth_frame(1, 2, 96, 64)
    dets = [[0.1, 0.1, 0.5, 0.5, 0.9], [0.3, 0.3, 0.9, 0.95, 0.8], [0.4, 0.2, 0.41, 0.9, 0.7]]
    out = oracle.draw_hollow_rects(rgb, dets, 96, 64)
    m = np.zeros(rgb.shape[:2], bool)
    for d in dets:
        r = oracle.rect_of_det(d, 96, 64)
        if r:
            m |= _outline(rgb.shape, *r)
    assert (out[m] == GREEN).all() and np.array_equal(out[~m], rgb[~m])
    jpeg = oracle.jpeg_encode_rgb(out, 95)
    item = oracle.stream_item(jpeg)
    assert item == b"--frame\r\nContent-Type: image/jpeg\r\n\r\n" + jpeg + b"\r\n\r\n"


def test_label_text_is_the_reference_format():
    # format!("{:.2}%", confidence * 100.0): f32 product, two decimals, correctly rounded
    for conf, text in ((0.9753, "97.53%"), (0.5000001, "50.00%"), (1.0, "100.00%"), (0.999999, "100.00%"), (0.12345, "12.35%"),
                       (0.125, "12.50%"), (0.51, "51.00%"), (0.07, "7.00%"), (0.999949, "99.99%")):
        assert oracle.label_text(np.float32(conf)) == text, conf
    rng = np.random.default_rng(2)
    for c in rng.uniform(0, 1, 2000).astype(np.float32):
        assert oracle.label_text(c) == "%.2f%%" % float(np.float32(c) * np.float32(100.0))  # (both round the exact f32 value)


def test_glyph_atlas_shapes_agree_with_freetype():
    """The coverage atlas (tools/make_glyph_atlas.py: rusttype / ab_glyph_rasterizer restated) cannot be pinned bit for
    bit -- those crates are not here -- but another rasteriser must agree on the shapes: FreeType (PIL) at the same size
    and sub-pixel position, total ink within 12 % and correlated."""
    from PIL import Image, ImageDraw, ImageFont

    path = "/usr/share/fonts/truetype/dejavu/DejaVuSansMono.ttf"
    if not os.path.exists(path):
        pytest.skip("no DejaVuSansMono on this box")
    atlas = np.load(os.path.join(ROOT, "tests", "golden", "glyph_atlas.npz"))
    font = ImageFont.truetype(path, 16 * 2048 / 2384)  # Scale 16 = ascent - descent = 2384 units of a 2048-unit em
    for key, ch in (("0_8", "8"), ("0_0", "0"), ("1_7", "7"), ("3_pct", "%"), ("2_dot", "."), ("5_4", "4")):
        a = atlas[key]
        x, y, w, h = (int(v) for v in a[:4])
        cov = a[4:].reshape(h, w)
        assert cov.min() >= 0.0 and cov.max() <= 1.0 and cov.max() > 0.5
        k = int(key.split("_")[0])
        im = Image.new("L", (96, 24), 0)
        ImageDraw.Draw(im).text((k * 16 * 1233 / 2384, 0), ch, fill=255, font=font)
        ft = np.asarray(im, np.float32)[y:y + h, x:x + w] / 255
        assert abs(ft.sum() - cov.sum()) <= 0.12 * cov.sum() + 0.5, (key, ft.sum(), cov.sum())
        if ch != ".":
            assert np.corrcoef(ft.ravel(), cov.ravel())[0, 1] > 0.6, key


def test_labels_blend_in_detection_order():
    rgb = synth.synth_frame(1, 2, 160, 120)
    a, b = [0.1, 0.1, 0.6, 0.6, 0.9753], [0.12, 0.13, 0.7, 0.7, 0.8801]
    ab, ba = oracle.draw_labels(rgb, [a, b], 160, 120), oracle.draw_labels(rgb, [b, a], 160, 120)
    assert not np.array_equal(ab, ba)  # b's rectangle cuts through a's label in one order, a's label covers it in the other
    # a lone label: rectangle pixels green, label pixels a blend towards green, everything else untouched
    one = oracle.draw_labels(rgb, [a], 160, 120)
    rect_only = oracle.draw_hollow_rects(rgb, [a], 160, 120)
    changed = (one != rect_only).any(2)
    ys, xs = np.nonzero(changed)
    l, t, _, _ = oracle.rect_of_det(a, 160, 120)
    assert len(ys) > 40 and xs.min() >= l and xs.max() < l + 60 and ys.min() >= t and ys.max() < t + 16
    assert (one[changed][:, 1].astype(int) >= rect_only[changed][:, 1].astype(int)).all()  # green never decreases
    # clipped at the frame border without touching anything outside
    edge = oracle.draw_labels(rgb, [[0.95, 0.92, 1.5, 1.5, 0.5]], 160, 120)
    assert edge.shape == rgb.shape


def test_annotate_encode_composite():
    W, H = 320, 240
    weights, priors = synth.synthetic_weights(), synth.gen_priors(W, H)
    jpeg = synth.encode_jpeg(synth.synth_frame(synth.DEFAULT_FRAME_SEED, 2, W, H))
    dets, out = oracle.annotate_encode_jpeg(jpeg, W, H, weights, priors, W, H)
    assert len(dets) > 0 and np.array_equal(dets, oracle.infer_jpeg(jpeg, W, H, weights, priors))
    frame = oracle.draw_labels(oracle.jpeg_decode_rgb(jpeg), dets, W, H)
    assert out == oracle.jpeg_encode_rgb(frame, 95)
