"""Row N1 on the GPU, through the C ABI: rectangles (inferer.rs:58-92) and the JPEG re-encode
(inferer.rs:39) against the CPU oracle, which is pinned byte for byte to libjpeg-turbo's own streams
(tests/test_oracle_encode.py).  Streams must be IDENTICAL; rectangle pixels exact."""
import os

import numpy as np
import pytest

import oracle
from helpers import assert_dets_match, dets_array
from infercam_onnx_amd import nn, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PICS = os.path.join(ROOT, "tests", "golden", "test_pics")


def _model(variant, weights, **kw):
    v = nn.UltrafaceVariant.W640H480 if variant == 640 else nn.UltrafaceVariant.W320H240
    W, H = v.width_height()
    return nn.UltrafaceModel(v, 0.5, 0.5, weights=weights, priors=synth.gen_priors(W, H), **kw)


@pytest.fixture(scope="module")
def model320(weights):
    with _model(320, weights, max_batch=8, max_src=(1280, 960), det_cap=512) as m:
        yield m


SIZES = [(640, 480), (320, 240), (1280, 720), (333, 217), (150, 100), (64, 52), (40, 24), (37, 29), (17, 9), (8, 8), (1, 1),
         (136, 8), (16, 200)]


@pytest.mark.parametrize("size", SIZES)
def test_encoder_streams_equal_the_oracle(model320, size):
    """MCU-aligned and ragged frames (replicated edge samples, dummy blocks), the reference's quality 95 (fast DCT)."""
    w, h = size
    rgb = synth.synth_frame(21, w * 7 + h, max(w, 16), max(h, 16))[:h, :w]
    assert model320.debug_encode_rgb(rgb, 95) == oracle.jpeg_encode_rgb(rgb, 95)


@pytest.mark.parametrize("quality", [96, 100, 75, 50, 10])
def test_encoder_other_qualities(model320, quality):
    """tjCompress2 switches to the accurate DCT from quality 96 on; other tables exercise the reciprocal quantiser."""
    rgb = synth.synth_frame(5, quality, 200, 120)
    assert model320.debug_encode_rgb(rgb, quality) == oracle.jpeg_encode_rgb(rgb, quality)


def test_encoder_hard_content(model320):
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, (96, 112, 3), dtype=np.uint8)  # saturated noise: 16-bit wrap-around, many 0xFF bytes
    chk = np.zeros((64, 64, 3), np.uint8)
    chk[::2, 1::2] = 255
    chk[1::2, ::2] = 255
    flat = np.full((48, 80, 3), 255, np.uint8)
    for rgb in (noise, chk, flat):
        for q in (95, 100):
            got = model320.debug_encode_rgb(rgb, q)
            assert got == oracle.jpeg_encode_rgb(rgb, q)
    # the stream really holds stuffed bytes
    assert b"\xff\x00" in model320.debug_encode_rgb(noise, 100)


def test_encoder_multipart_framing(model320):
    rgb = synth.synth_frame(5, 5, 96, 64)
    jpeg = oracle.jpeg_encode_rgb(rgb, 95)
    assert model320.debug_encode_rgb(rgb, 95, multipart=True) == oracle.stream_item(jpeg)
    assert model320.debug_encode_rgb(rgb, 95) == jpeg  # and back to the plain header


def test_rectangles_and_labels_equal_the_oracle(model320):
    rng = np.random.default_rng(11)
    rgb = synth.synth_frame(1, 3, 320, 240)
    dets = np.concatenate([
        rng.uniform(-0.3, 1.3, (40, 5)).astype(np.float32),                       # any order of corners, off-frame
        np.array([[0.0, 0.0, 1.0, 1.0, 0.9], [0.5, 0.5, 0.5, 0.9, 0.8],           # whole frame; zero width
                  [0.25, 0.25, 0.2534, 0.75, 0.7], [np.nan, 0.1, 0.5, 0.5, 0.6],   # one pixel wide; NaN corner
                  [-5.0, -5.0, 6.0, 6.0, 0.5], [0.999, 0.999, 1.5, 1.5, 0.5]], np.float32)])
    dets[:, 4] = np.abs(dets[:, 4]) % 1.0  # confidences in [0, 1): labels "0.00%" .. "99.99%"
    dets[3, 4], dets[4, 4] = 1.0, 0.999999  # "100.00%" twice (seven characters)
    for label in ((320, 240), (1280, 720), (100.5, 77.25)):
        got = model320.debug_draw_labels(rgb, dets, label, text=False)
        assert np.array_equal(got, oracle.draw_hollow_rects(rgb, dets, *label)), label
        # with the labels: order matters (a label blends over earlier rectangles, later rectangles cut through it)
        got = model320.debug_draw_labels(rgb, dets, label)
        assert np.array_equal(got, oracle.draw_labels(rgb, dets, *label)), label
    assert np.array_equal(model320.debug_draw_labels(rgb, np.zeros((0, 5), np.float32), (320, 240)), rgb)


def test_product_label_glyphs_agree_with_freetype(model320):
    """The label coverage table compiled into the PRODUCT (csrc/glyph_atlas.inc; restated from rusttype /
    ab_glyph_rasterizer, parity with those crates unpinned -- DESIGN.md section 2) drawn by the GPU on a black frame,
    against FreeType (PIL) rendering the same strings of inferer.rs:80-88 at the same origin and scale: ink bounding
    box within one pixel per side, total ink within 12 %, correlation > 0.8 -- so a regenerated table cannot drift
    silently.  (UFD_ANNOT_NO_TEXT stays the mode whose pixels are exact.)"""
    from PIL import Image, ImageDraw, ImageFont

    path = "/usr/share/fonts/truetype/dejavu/DejaVuSansMono.ttf"
    if not os.path.exists(path):
        pytest.skip("no DejaVuSansMono on this box")
    font = ImageFont.truetype(path, 16 * 2048 / 2384)  # Scale 16 = ascent - descent = 2384 units of a 2048-unit em
    W, H, x0, y0 = 160, 64, 20, 12
    black = np.zeros((H, W, 3), np.uint8)
    seen = set()
    for conf in (0.1234, 0.5678, 0.9012, 0.3456, 0.7899, 1.0, 0.0705):
        text = oracle.label_text(np.float32(conf))
        seen |= set(text)
        det = np.array([[x0 / W, y0 / H, 0.95, 0.95, conf]], np.float32)
        with_text = model320.debug_draw_labels(black, det, (W, H)).astype(np.float32)
        rect_only = model320.debug_draw_labels(black, det, (W, H), text=False).astype(np.float32)
        cov = (with_text - rect_only)[:, :, 1] / 255.0  # colour (0, 255, 0) blended over black: green = 255 * coverage
        assert (with_text[:, :, 0] == 0).all() and (with_text[:, :, 2] == 0).all()
        im = Image.new("L", (W, H), 0)
        ImageDraw.Draw(im).text((x0, y0), text, fill=255, font=font)
        ft = np.asarray(im, np.float32) / 255.0
        for a in (cov, ft):  # the rectangle's own top row / left column are already green: not comparable
            a[y0, :] = 0
            a[:, x0] = 0

        def ink_box(a):
            ys, xs = np.nonzero(a > 0.25)
            return np.array([xs.min(), ys.min(), xs.max(), ys.max()])

        assert np.abs(ink_box(cov) - ink_box(ft)).max() <= 1, (text, ink_box(cov), ink_box(ft))
        assert abs(cov.sum() - ft.sum()) <= 0.12 * ft.sum(), (text, cov.sum(), ft.sum())
        assert np.corrcoef(cov.ravel(), ft.ravel())[0, 1] > 0.8, text
    assert seen == set("0123456789.%")


def test_many_overlapping_labels_keep_the_reference_order(model320):
    """Hundreds of detections piled on the same spot (more than one 256-operation pass of the tile kernel): every pixel
    must see rectangles and label blends in detection order."""
    rng = np.random.default_rng(5)
    rgb = synth.synth_frame(1, 4, 320, 240)
    n = 700
    dets = np.zeros((n, 5), np.float32)
    dets[:, 0] = 0.3 + rng.uniform(-0.05, 0.05, n)
    dets[:, 1] = 0.3 + rng.uniform(-0.05, 0.05, n)
    dets[:, 2] = dets[:, 0] + rng.uniform(0.05, 0.4, n)
    dets[:, 3] = dets[:, 1] + rng.uniform(0.05, 0.4, n)
    dets[:, 4] = rng.uniform(0.5, 1.0, n)
    assert np.array_equal(model320.debug_draw_labels(rgb, dets, (320, 240)), oracle.draw_labels(rgb, dets, 320, 240))


def _expected_stream(jpeg, dets, label, quality=95):
    """The oracle's draw + encode of the oracle's decode, with the detections the GPU reported (detections are
    compared with the oracle's separately: a rectangle corner may sit on an integer boundary)."""
    frame = oracle.draw_labels(oracle.jpeg_decode_rgb(jpeg), dets_array(dets), *label)
    return oracle.jpeg_encode_rgb(frame, quality)


def test_annotate_batch_end_to_end_640(weights):
    """C3's workload through ufd_annotate_jpeg_batch: fused 4:2:0 path, label size = frame size and the router's 1280x720."""
    W, H = 640, 480
    jpegs = synth.synth_jpeg_pool(0, 8, W, H, quality=90, subsampling="4:2:0")
    priors = synth.gen_priors(W, H)
    with _model(640, weights, max_batch=8, max_src=(W, H), det_cap=1024) as m:
        for label in ((W, H), (1280, 720)):
            dets, status, streams = m.annotate_jpeg_batch(jpegs, label)
            assert status == [0] * 8
            for j, d, s in zip(jpegs, dets, streams):
                assert_dets_match(dets_array(d), oracle.infer_jpeg(j, W, H, weights, priors), what="annotate")
                assert s == _expected_stream(j, d, label)
                assert oracle.jpeg_decode_rgb(s).shape == (H, W, 3)


def test_annotate_pipelined_with_a_corrupt_frame(weights):
    """Six annotate batches in flight over the handle's contexts; a corrupt slot yields status DECODE and no stream,
    the other frames of its batch are unaffected (inferer.rs:35-36 would panic the task)."""
    W, H = 320, 240
    pool = synth.synth_jpeg_pool(3, 24, W, H, quality=90, subsampling="4:2:0")
    pool[5] = pool[5][:200] + bytes(50) + pool[5][260:len(pool[5]) // 2]
    with _model(320, weights, max_batch=4, max_src=(W, H), det_cap=512) as m:
        batches = [m.prep_annotate_batch(pool[i * 4:(i + 1) * 4], (1280, 720), multipart=True) for i in range(6)]
        for rnd in range(2):
            tickets = [m.submit_annotate_batch(b) for b in batches]
            for bi, t in enumerate(tickets):
                dets, status, streams = m.wait(t)
                for k in range(4):
                    j = pool[bi * 4 + k]
                    if bi * 4 + k == 5:
                        assert status[k] == nn.UFD_E_DECODE and streams[k] is None
                        continue
                    assert status[k] == 0
                    assert streams[k] == oracle.stream_item(_expected_stream(j, dets[k], (1280, 720)))


def test_annotate_server_default_1280x720_to_320(weights):
    """The reference server's operating point: 1280x720 camera frames, UltraFace-320, rectangles on the full frame."""
    jpegs = synth.synth_jpeg_pool(1, 3, 1280, 720, quality=90, subsampling="4:2:0")
    jpegs.append(synth.encode_jpeg(synth.synth_frame(9, 9, 1280, 720), quality=85, subsampling="4:2:2"))
    with _model(320, weights, max_batch=4, max_src=(1280, 720), det_cap=512) as m:
        dets, status, streams = m.annotate_jpeg_batch(jpegs, (1280, 720))
        assert status == [0] * 4
        for j, d, s in zip(jpegs, dets, streams):
            assert s == _expected_stream(j, d, (1280, 720))


@pytest.mark.parametrize("name", sorted(os.listdir(PICS)) if os.path.isdir(PICS) else [])
def test_annotate_reference_test_pictures(model320, name):
    """The reference's own fixtures (progressive, 640 x {427..960}: ragged MCU rows, a dummy block row at 676)."""
    jpeg = open(os.path.join(PICS, name), "rb").read()
    d, s = model320.annotate_jpeg(jpeg, (1280, 720))
    assert s == _expected_stream(jpeg, d, (1280, 720))


def test_annotate_without_text_flag(weights):
    W, H = 320, 240
    jpegs = synth.synth_jpeg_pool(2, 2, W, H, quality=90, subsampling="4:2:0")
    with _model(320, weights, max_batch=2, max_src=(W, H), det_cap=4420) as m:
        dets, status, streams = m.annotate_jpeg_batch(jpegs, (W, H), text=False)
        for j, d, s in zip(jpegs, dets, streams):
            frame = oracle.draw_hollow_rects(oracle.jpeg_decode_rgb(j), dets_array(d), W, H)
            assert s == oracle.jpeg_encode_rgb(frame, 95)


def test_annotate_streams_into_pinned_and_into_ordinary_memory(weights):
    """The finished streams reach a PINNED caller buffer (ufd_host_alloc) by a launch at the end of the batch's own chain
    (k_fetch_streams: nothing left to copy in ufd_wait); an ordinary buffer is filled by a copy in ufd_wait.  Same bytes,
    with six batches in flight and one frame per batch corrupt."""
    from infercam_onnx_amd import synth

    W, H = 640, 480
    with _model(640, weights, max_batch=4, max_src=(W, H), det_cap=512, profile=True) as m:
        sets = []
        for k in range(6):
            js = [synth.encode_jpeg(synth.synth_frame(61, 4 * k + i, W, H), subsampling="4:2:0", quality=90) for i in range(4)]
            js[k % 4] = js[k % 4][: len(js[k % 4]) // 3]
            sets.append(js)
        got = {}
        for pinned in (True, False):
            bs = [m.prep_annotate_batch(js, (1280, 720), pinned=pinned) for js in sets]
            m.profile_reset()
            tickets = [m.submit_annotate_batch(b) for b in bs]
            got[pinned] = [m.wait(t) for t in tickets]
            launched = {p["name"] for p in m.profile_read() if p["launches"]}
            assert ("d2h_streams" in launched) == pinned, (pinned, sorted(launched))  # the path taken is the one meant
        for k in range(6):
            (d0, s0, j0), (d1, s1, j1) = got[True][k], got[False][k]
            assert s0 == s1 and s0[k % 4] != 0 and d0 == d1 and j0 == j1
            assert j0[k % 4] is None and all(j is not None and j[:2] == b"\xff\xd8" for i, j in enumerate(j0) if i != k % 4)


def test_annotate_pinned_buffer_of_exactly_the_packed_size(weights):
    """A pinned output buffer whose size is the packed total of the batch's streams and no multiple of 16: the launch that
    writes the streams (k_fetch_streams) moves whole 16-byte pieces, so the last stream's ragged tail -- its EOI marker --
    comes by a copy of its own in ufd_wait (advisor finding, round 4: it used to be lost with the frame reported UFD_OK)."""
    W, H = 320, 240
    with _model(320, weights, max_batch=2, max_src=(W, H), det_cap=512, profile=True) as m:
        for seed in range(8):
            jpegs = synth.synth_jpeg_pool(40 + seed, 2, W, H, quality=90, subsampling="4:2:0")
            ref_d, ref_st, ref_s = m.annotate_jpeg_batch(jpegs, (W, H))
            total = ((len(ref_s[0]) + 15) & ~15) + len(ref_s[1])  # streams are packed at 16-byte boundaries
            if total % 16:
                break
        assert total % 16, "no batch with a ragged packed size among the seeds"
        b = m.prep_annotate_batch(jpegs, (W, H), cap_bytes=total, pinned=True)
        m.profile_reset()
        dets, status, streams = m.wait(m.submit_annotate_batch(b))
        assert "d2h_streams" in {p["name"] for p in m.profile_read() if p["launches"]}  # the pinned path is the one tested
        assert status == [0, 0] and streams == ref_s and streams[1][-2:] == b"\xff\xd9" and dets == ref_d
        b = m.prep_annotate_batch(jpegs, (W, H), cap_bytes=total - 1, pinned=True)  # one byte short: the last frame is truncated
        dets, status, streams = m.wait(m.submit_annotate_batch(b))
        assert status == [0, nn.UFD_E_TRUNCATED] and streams[0] == ref_s[0] and streams[1] is None


def test_annotate_output_buffer_too_small(weights):
    W, H = 320, 240
    jpegs = synth.synth_jpeg_pool(2, 4, W, H, quality=90, subsampling="4:2:0")
    with _model(320, weights, max_batch=4, max_src=(W, H), det_cap=512) as m:
        ref_d, ref_st, ref_s = m.annotate_jpeg_batch(jpegs, (W, H))
        b = m.prep_annotate_batch(jpegs, (W, H), out_bytes_per_frame=(len(ref_s[0]) + len(ref_s[1]) + 64) // 4)
        dets, status, streams = m.wait(m.submit_annotate_batch(b))
        assert streams[0] == ref_s[0] and streams[1] == ref_s[1]
        assert status[:2] == [0, 0] and status[2:] == [nn.UFD_E_TRUNCATED] * 2 and streams[2] is None and streams[3] is None
        assert dets[2] == ref_d[2]  # detections of a truncated frame are still reported


def test_annotate_mixed_frame_sizes_and_flavours_in_one_batch(weights):
    """Frames of different sizes, subsamplings and coding modes in one annotate batch (the encoder kernels take every
    frame's own dimensions; progressive and 4:4:4 frames go through the host entropy decoder and the generic upsampler):
    every stream equals the oracle's for that frame."""
    frames = [
        synth.encode_jpeg(synth.synth_frame(31, 0, 320, 240)),
        synth.encode_jpeg(synth.synth_frame(31, 1, 640, 480), subsampling="4:2:2"),
        synth.encode_jpeg(synth.synth_frame(31, 2, 72, 56), subsampling="4:4:4"),
        synth.encode_jpeg(synth.synth_frame(31, 3, 333, 217)),
        synth.encode_jpeg(synth.synth_frame(31, 4, 640, 360), progressive=True),
        synth.encode_jpeg(synth.synth_frame(31, 5, 24, 24), restart_rows=1),
    ]
    with _model(320, weights, max_batch=8, max_src=(640, 480), det_cap=4420) as m:
        for label in ((1280, 720), (320, 240)):
            dets, status, streams = m.annotate_jpeg_batch(frames, label)
            assert status == [0] * len(frames)
            for j, d, s in zip(frames, dets, streams):
                assert s == _expected_stream(j, d, label)


def test_annotate_frames_of_the_other_libjpeg_layouts(weights):
    """Round 6: 4:1:1, 4:1:0, 4:4:0 and RGB-colourspace camera frames (libjpeg-turbo-written fixtures) through the whole
    `Inferer::run` iteration -- decode (generic upsampler), detect, draw, re-encode: streams equal the oracle's."""
    from test_oracle_jpeg import layout_fixtures

    fx = layout_fixtures()
    frames = [fx["411_640x480"][0]["base"], fx["410_640x480"][0]["prog"], fx["440_640x480"][0]["dri_row"],
              fx["rgb_adobe0_320x240"][0]["base"], fx["411_321x243"][0]["base"], fx["y11c22_150x100"][0]["base"]]
    priors = synth.gen_priors(320, 240)
    with _model(320, weights, max_batch=8, max_src=(640, 480), det_cap=4420) as m:
        dets, status, streams = m.annotate_jpeg_batch(frames, (1280, 720))
        assert status == [0] * len(frames)
        for j, d, s in zip(frames, dets, streams):
            assert_dets_match(dets_array(d), oracle.infer_jpeg(j, 320, 240, weights, priors), what="annotate layouts")
            assert s == _expected_stream(j, d, (1280, 720))
