"""Oracle pinning, row A1: the CPU JPEG restatement is bit-exact against libjpeg-turbo (the
library behind the reference's turbojpeg crate) on committed fixtures and on live PIL decodes."""
import hashlib
import io
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


def test_small_fixtures_bit_exact(oracle_lib):
    z = np.load(os.path.join(G, "jpeg_small.npz"))
    names = sorted(k[:-5] for k in z.files if k.endswith("_jpeg"))
    assert len(names) == 5
    for n in names:
        got = oracle_lib.jpeg_decode_rgb(z[n + "_jpeg"].tobytes())
        assert np.array_equal(got, z[n + "_rgb"]), n


def test_reference_test_pics_progressive(oracle_lib):
    """The reference's own test pictures (integration_tests.rs:20-29) are progressive 4:2:0."""
    meta = json.load(open(os.path.join(G, "test_pics.json")))
    assert len(meta) == 8
    for f, m in meta.items():
        b = open(os.path.join(G, "test_pics", f), "rb").read()
        info = oracle_lib.jpeg_probe(b)
        assert info.progressive == 1 and [info.height, info.width, 3] == m["shape"]
        rgb = oracle_lib.jpeg_decode_rgb(b)
        assert hashlib.sha256(rgb.tobytes()).hexdigest() == m["sha256"], f


@pytest.mark.parametrize("size", [(640, 480), (641, 479), (17, 9), (2, 2), (1, 1), (100, 37)])
@pytest.mark.parametrize("subsampling", ["4:2:0", "4:2:2", "4:4:4"])
def test_against_live_libjpeg_turbo(oracle_lib, size, subsampling):
    from PIL import Image
    from infercam_onnx_amd import synth

    w, h = size
    for kw in ({}, {"restart_rows": 1}, {"progressive": True}, {"optimize": True}, {"quality": 30}, {"quality": 100}):
        jpeg = synth.encode_jpeg(synth.synth_frame(3, w * h, w, h), subsampling=subsampling, **kw)
        ref = np.asarray(Image.open(io.BytesIO(jpeg)).convert("RGB"))
        assert np.array_equal(oracle_lib.jpeg_decode_rgb(jpeg), ref), (size, subsampling, kw)


def test_corrupt_streams_are_errors(oracle_lib):
    from infercam_onnx_amd import synth

    jpeg = synth.encode_jpeg(synth.synth_frame(3, 0, 64, 48))
    for bad in (jpeg[:200], jpeg[:-2], b"\xff\xd8\xff", b"", b"not a jpeg at all"):
        with pytest.raises(oracle_lib.OracleError):
            oracle_lib.jpeg_decode_rgb(bad)


@pytest.mark.parametrize("subsampling", ["4:2:0", "4:2:2"])
def test_mjpg_without_dht(oracle_lib, subsampling):
    """camera MJPG may omit DHT => the Annex-K default tables.  A non-optimised PIL stream carries exactly
    those tables; without its DHT segments (a) libjpeg-turbo itself (PIL), (b) the oracle and (c) the product's host
    entropy decoder must all give what the ORIGINAL stream gives (identical tables => identical pixels / coefficients)."""
    from PIL import Image
    from infercam_onnx_amd import nn, synth

    for kw in ({}, {"restart_rows": 1}):
        jpeg = synth.encode_jpeg(synth.synth_frame(9, 4, 320, 240), subsampling=subsampling, **kw)
        bare = synth.strip_dht(jpeg)
        assert len(bare) < len(jpeg) and b"\xff\xc4" not in bare[:bare.index(b"\xff\xda")]
        ref = np.asarray(Image.open(io.BytesIO(jpeg)).convert("RGB"))
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(bare)).convert("RGB")), ref)  # libjpeg-turbo's own default tables
        assert np.array_equal(oracle_lib.jpeg_decode_rgb(bare), ref)
        c0, w0, h0 = nn.jpeg_coefficients(jpeg)
        c1, w1, h1 = nn.jpeg_coefficients(bare)
        assert (w0, h0) == (w1, h1) == (320, 240) and np.array_equal(c0, c1)


# ---------------------------------------------------------------- every layout libjpeg-turbo decodes (tests/golden/jpeg_layouts.npz)
def layout_fixtures():
    """{name: ({kind: jpeg bytes}, rgb array or None, sha256 hex or None)} from tools/make_layout_golden.py: streams libjpeg-turbo
    2.1.2 WROTE (sampling factors / colour space set in comp_info: 4:4:0, 4:1:1, 4:1:0, 4:4:1, 3x, chroma finer than luma,
    RGB colour space ...) and the pixels libjpeg-turbo 3.1.x DECODES them to."""
    z = np.load(os.path.join(G, "jpeg_layouts.npz"))
    out = {}
    for k in z.files:
        name, kind = k.split("/")
        e = out.setdefault(name, [{}, None, None])
        if kind == "rgb":
            e[1] = z[k]
        elif kind == "sha256":
            e[2] = z[k].tobytes().hex()
        else:
            e[0][kind] = z[k].tobytes()
    return out


N_LAYOUT_STREAMS = 329


def check_pixels(name, kind, got, rgb, sha):
    if rgb is not None:
        assert got.shape == rgb.shape and np.array_equal(got, rgb), "%s/%s: %d samples differ" % (name, kind, (got != rgb).sum())
    else:
        assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == sha, "%s/%s" % (name, kind)


def test_every_libjpeg_layout_bit_exact(oracle_lib):
    """4:4:0 (h1v2 fancy), 4:1:1 / 4:1:0 / 4:4:1 and the other integral expansions (plain replication, jdsample.c
    int_upsample), luma coarser than chroma, RGB-colourspace streams by libjpeg's marker rules -- as baseline, with two
    restart layouts, progressive, with optimised tables and non-interleaved (a scan per component): the oracle gives
    libjpeg-turbo's pixels on all of the streams."""
    fx = layout_fixtures()
    assert len(fx) == 53 and sum(len(v[0]) for v in fx.values()) == N_LAYOUT_STREAMS
    for name, (streams, rgb, sha) in sorted(fx.items()):
        for kind, jpeg in streams.items():
            check_pixels(name, kind, oracle_lib.jpeg_decode_rgb(jpeg), rgb, sha)


def test_product_host_decoder_takes_every_layout():
    """The product's host entropy decoder (no GPU needed) no longer refuses expansion factors above 2: every fixture
    stream parses, and the five entropy codings of one layout give the SAME coefficient slab (that is what makes their
    pixels equal)."""
    from infercam_onnx_amd import nn

    for name, (streams, _, _) in sorted(layout_fixtures().items()):
        ref = None
        for kind, jpeg in streams.items():
            coef, w, h = nn.jpeg_coefficients(jpeg)
            assert "%dx%d" % (w, h) == name.rsplit("_", 1)[1]
            if kind.startswith("nonint"):
                continue  # (a component's own scan does not code the MCU-padding blocks: those stay zero in the slab)
            assert ref is None or np.array_equal(coef, ref), "%s/%s" % (name, kind)
            ref = coef


def _patch_sof(jpeg, comp, hv):
    i = 2
    while jpeg[i + 1] not in (0xC0, 0xC1, 0xC2):
        i += 2 + ((jpeg[i + 2] << 8) | jpeg[i + 3])
    b = bytearray(jpeg)
    b[i + 11 + 3 * comp] = hv
    return bytes(b)


def test_layouts_libjpeg_refuses_are_errors(oracle_lib):
    """Fractional expansion (hmax not a multiple of a component's factor: jdsample.c JERR_FRACT_SAMPLE_NOTIMPL) and more
    than 10 blocks in an interleaved MCU (jdinput.c JERR_BAD_MCU_SIZE) fail in libjpeg-turbo, so `decompress_image`'s
    `expect` panics (inferer.rs:35-36): the oracle and the product's host decoder return an error, never pixels."""
    from PIL import Image
    from infercam_onnx_amd import nn, synth

    jpeg = synth.encode_jpeg(synth.synth_frame(3, 0, 64, 48), subsampling="4:4:4")
    for bad in (_patch_sof(_patch_sof(jpeg, 0, 0x31), 1, 0x21),   # hmax 3, chroma 2: fractional
                _patch_sof(_patch_sof(jpeg, 0, 0x42), 1, 0x21)):  # 8 + 2 + 1 = 11 blocks
        with pytest.raises(Exception):
            Image.open(io.BytesIO(bad)).convert("RGB")
        with pytest.raises(oracle_lib.OracleError):
            oracle_lib.jpeg_decode_rgb(bad)
        with pytest.raises(nn.UfdError):
            nn.jpeg_coefficients(bad)
