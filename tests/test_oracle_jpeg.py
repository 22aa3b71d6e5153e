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
