#!/usr/bin/env python3
"""Golden vectors for row A1 (turbojpeg::decompress_image, inferer.rs:35) over every component layout libjpeg-turbo
decodes and PIL cannot write: JPEG streams written by libjpeg-turbo itself (the system libjpeg.so.8 = libjpeg-turbo
2.1.2, through tools/make_encode_golden.py's ctypes TurboEncoder with the sampling factors / colour space set in
comp_info) and the pixels libjpeg-turbo decodes them to (PIL's bundled 3.1.x: jpeg_read_header defaults = ISLOW IDCT +
fancy upsampling = tjDecompress2(flags = 0)).

Layouts: 4:4:0, 4:1:1, 4:1:0, 4:4:1 and the less common legal ones (luma 3x1 / 1x3 / 3x2 / 2x4, chroma finer than luma,
chroma at half the luma rate of a 4-wide MCU, a different rate per chroma plane), RGB-colourspace streams (component
ids R G B with an Adobe transform-0 marker, with no marker at all, with a JFIF marker that overrides the ids, subsampled),
YCbCr with an Adobe transform-1 marker or no marker, a one-component stream whose SOF says 2x2.  Each as baseline, with
restart intervals (one MCU row; 3 MCUs), progressive, with optimised tables, and NON-INTERLEAVED (one sequential scan per
component; luma alone + the chroma planes together, with a restart interval): the entropy coding differs, the coefficients
of the visible blocks and so the pixels do not -- one pixel array per (layout, size).

Run in the build container:  python tools/make_layout_golden.py   ->  tests/golden/jpeg_layouts.npz
"""
import hashlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from make_encode_golden import JCS_RGB, JDCT_ISLOW, TurboEncoder  # noqa: E402

Y, C = (1, 1), (1, 1)
LAYOUTS = {
    "440": [(1, 2), C, C],            # h1v2: the one fancy upsampler PIL never exercises
    "411": [(4, 1), C, C],            # DV / some UVC bridges: h4v1 = plain replication (jdsample.c int_upsample)
    "410": [(4, 2), C, C],            # 10 blocks per MCU, the limit of T.81
    "441": [(1, 4), C, C],
    "y31": [(3, 1), C, C],
    "y13": [(1, 3), C, C],
    "y32": [(3, 2), C, C],
    "y24": [(2, 4), C, C],
    "y41c21": [(4, 1), (2, 1), (2, 1)],   # chroma h2v1 (fancy) inside a 4-wide MCU
    "y22c21": [(2, 2), (2, 1), (2, 1)],   # chroma h1v2 (fancy) inside a 2x2 MCU
    "y22c12": [(2, 2), (1, 2), (1, 2)],   # chroma h2v1 (fancy) inside a 2x2 MCU
    "y11c22": [(1, 1), (2, 2), (2, 2)],   # LUMA upsampled h2v2 (fancy), chroma at full rate
    "y11c21": [(1, 1), (2, 1), (2, 1)],
    "y21c12": [(2, 1), (1, 2), (1, 2)],   # luma h1v2, chroma h2v1
    "cb11cr22": [(2, 2), (1, 1), (2, 2)], # one chroma plane subsampled, the other not
}
COLOURS = {
    "rgb_adobe0": dict(colorspace=JCS_RGB),                                     # ids R G B, Adobe transform 0, no JFIF
    "rgb_ids_only": dict(colorspace=JCS_RGB, write_adobe=False),                # ids R G B decide
    "rgb_ids_jfif": dict(colorspace=JCS_RGB, write_adobe=False, write_jfif=True),  # JFIF => YCbCr whatever the ids say
    "rgb_420": dict(colorspace=JCS_RGB, samp=[(2, 2), C, C]),                   # subsampled G / B planes
    "ycc_no_marker": dict(write_jfif=False),                                    # ids 1 2 3, no marker => YCbCr
    "ycc_adobe1": dict(write_jfif=False, write_adobe=True),                     # Adobe transform 1 => YCbCr
}
KINDS = {"base": {}, "dri_row": {"restart_rows": 1}, "dri_3": {"restart_mcus": 3}, "prog": {"progressive": True},
         "opt": {"optimize": True},
         # NON-INTERLEAVED sequential files: one scan per component (its blocks are ceil(w / 8) x ceil(h / 8) of the component, not
         # MCU-padded), and luma alone + the two chroma planes interleaved; the second with a restart interval
         "nonint": {"scans": [((0,), 0, 63, 0, 0), ((1,), 0, 63, 0, 0), ((2,), 0, 63, 0, 0)]},
         "nonint_y_cc_dri": {"scans": [((0,), 0, 63, 0, 0), ((1, 2), 0, 63, 0, 0)], "restart_mcus": 3}}
SIZES_ALL = [(150, 100), (37, 29)]
SIZES_FEW = {(5, 3): ("440", "411", "410", "y11c22"), (640, 480): ("440", "411", "410"), (321, 243): ("411", "440", "y22c21")}
PIXEL_LIMIT = 5000   # larger frames are committed as the sha256 of their pixels


def pil_decode(jpeg):
    from PIL import Image

    return np.asarray(Image.open(io.BytesIO(jpeg)).convert("RGB"))


def sof_sampling(jpeg):
    i = 2
    while jpeg[i + 1] not in (0xC0, 0xC1, 0xC2):
        i += 2 + ((jpeg[i + 2] << 8) | jpeg[i + 3])
    n = jpeg[i + 9]
    return [(jpeg[i + 11 + 3 * c] >> 4, jpeg[i + 11 + 3 * c] & 15) for c in range(n)], i


def main():
    from infercam_onnx_amd import synth

    enc = TurboEncoder()
    out = {}

    def add(name, rgb, w, h, **kw):
        ref = None
        for kind, kkw in KINDS.items():
            if w * h > 20000 and kind in ("dri_3", "opt", "nonint_y_cc_dri"):
                continue  # (large frames: four kinds are enough)
            jpeg = enc.encode(rgb, 85, JDCT_ISLOW, **kw, **kkw)
            px = pil_decode(jpeg)
            assert px.shape == (h, w, 3)
            assert ref is None or np.array_equal(px, ref), (name, kind)  # same coefficients, same pixels
            ref = px
            out["%s/%s" % (name, kind)] = np.frombuffer(jpeg, np.uint8)
        if w * h <= PIXEL_LIMIT:
            out[name + "/rgb"] = ref
        else:
            out[name + "/sha256"] = np.frombuffer(hashlib.sha256(ref.tobytes()).digest(), np.uint8)
        return jpeg

    for (w, h) in SIZES_ALL + list(SIZES_FEW):
        rgb = synth.synth_frame(21, w + h, w, h)
        for lname, samp in LAYOUTS.items():
            if (w, h) in SIZES_FEW and lname not in SIZES_FEW[(w, h)]:
                continue
            j = add("%s_%dx%d" % (lname, w, h), rgb, w, h, samp=samp)
            assert sof_sampling(j)[0] == samp
    for (w, h) in ((150, 100), (320, 240)):
        rgb = synth.synth_frame(22, w, w, h)
        for cname, kw in COLOURS.items():
            add("%s_%dx%d" % (cname, w, h), rgb, w, h, **kw)
    # one component whose SOF claims 2x2: a single-component scan is never interleaved, the MCU is one block
    from PIL import Image

    gray = synth.synth_frame(23, 0, 67, 45)[:, :, 1]
    b = io.BytesIO()
    Image.fromarray(gray).save(b, "JPEG", quality=85)
    g = bytearray(b.getvalue())
    (s,), i = sof_sampling(bytes(g))
    assert s == (1, 1)
    ref = pil_decode(bytes(g))
    g[i + 11] = 0x22
    assert np.array_equal(pil_decode(bytes(g)), ref)
    out["gray_sof22_67x45/base"] = np.frombuffer(bytes(g), np.uint8)
    out["gray_sof22_67x45/rgb"] = ref

    path = os.path.join(ROOT, "tests", "golden", "jpeg_layouts.npz")
    np.savez_compressed(path, **out)
    names = sorted({k.split("/")[0] for k in out})
    print("wrote %s: %d bytes, %d layouts x sizes, %d streams" % (path, os.path.getsize(path), len(names),
                                                                    sum(1 for k in out if not k.endswith(("/rgb", "/sha256")))))


if __name__ == "__main__":
    main()
