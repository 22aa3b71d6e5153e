#!/usr/bin/env python3
"""Golden vectors for row N1 (annotate + JPEG re-encode, inferer.rs:38-40): RGB frames and the JPEG
bytes libjpeg-turbo itself produces for them with the settings turbojpeg::compress_image(&frame, 95,
Subsamp::Sub2x2) selects (tjCompress2, flags 0: quality 95, 4:2:0, Annex-K Huffman tables, no restart
markers, JFIF header and -- below quality 96 -- the FAST integer forward DCT, JDCT_IFAST).

The reference's encoder lives in turbojpeg 0.5.2 / turbojpeg-sys 0.2.2 (Cargo.lock:2617-2640), which
builds libjpeg-turbo 2.1.x from source; neither is under /root/reference.  This script drives the
libjpeg-turbo 2.1.2 that this image ships as /usr/lib/x86_64-linux-gnu/libjpeg.so.8 (the same library
family and major version, x86-64 SIMD build) through its public libjpeg API with ctypes -- there are no
headers in the image, so the three struct offsets it needs are checked against values the library
itself writes (struct size, input_gamma, the JFIF defaults) before anything is encoded.  PIL's bundled
libjpeg-turbo 3.1.x is used as a second opinion for the accurate-DCT (JDCT_ISLOW) variant, which PIL
can select and the two builds must agree on byte for byte.

Run in the build container:  python tools/make_encode_golden.py   ->  tests/golden/encode_q95_420.npz
"""
import ctypes
import io
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LIBJPEG = "/usr/lib/x86_64-linux-gnu/libjpeg.so.8"
JPEG_LIB_VERSION = 80
STRUCT_SIZE = 584        # sizeof(struct jpeg_compress_struct), v8 ABI, LP64 (jpeg_CreateCompress checks it)
OFF_IMAGE_WIDTH = 48     # image_width, image_height, input_components, in_color_space
OFF_INPUT_GAMMA = 64
OFF_COMP_INFO = 104
OFF_NUM_COMPONENTS = 92
OFF_NUM_SCANS = 272        # int num_scans; const jpeg_scan_info *scan_info at 280
OFF_OPTIMIZE_CODING = 296
OFF_DCT_METHOD = 312
OFF_RESTART_INTERVAL = 316   # unsigned restart_interval (MCUs), int restart_in_rows
OFF_WRITE_JFIF = 324
OFF_WRITE_ADOBE = 336
JCS_GRAYSCALE, JCS_RGB, JCS_YCBCR = 1, 2, 3
JDCT_ISLOW, JDCT_IFAST = 0, 1


class TurboEncoder:
    def __init__(self):
        L = ctypes.CDLL(LIBJPEG)
        vp = ctypes.c_void_p
        L.jpeg_std_error.restype = vp
        L.jpeg_std_error.argtypes = [vp]
        L.jpeg_CreateCompress.argtypes = [vp, ctypes.c_int, ctypes.c_size_t]
        L.jpeg_mem_dest.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_ulong)]
        for f in ("jpeg_set_defaults", "jpeg_finish_compress", "jpeg_destroy_compress"):
            getattr(L, f).argtypes = [vp]
        L.jpeg_set_quality.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        L.jpeg_set_colorspace.argtypes = [vp, ctypes.c_int]
        L.jpeg_simple_progression.argtypes = [vp]
        L.jpeg_start_compress.argtypes = [vp, ctypes.c_int]
        L.jpeg_write_scanlines.argtypes = [vp, vp, ctypes.c_uint]
        L.jpeg_write_scanlines.restype = ctypes.c_uint
        self.L = L

    def encode(self, rgb, quality=95, dct=JDCT_IFAST, samp=None, colorspace=None, restart_mcus=0, restart_rows=0,
               progressive=False, optimize=False, write_jfif=None, write_adobe=None, scans=None):
        """samp: [(h, v)] * 3 sampling factors written into comp_info (any layout libjpeg accepts, not only tjCompress2's
        five); colorspace: JCS_* of the stream (None = YCbCr; JCS_RGB writes component ids R, G, B and an Adobe marker
        with transform 0 instead of JFIF); write_jfif / write_adobe override which of the two markers is written."""
        L = self.L
        h, w, _ = rgb.shape
        rgb = np.ascontiguousarray(rgb, np.uint8)
        err = ctypes.create_string_buffer(1024)
        cinfo = ctypes.create_string_buffer(STRUCT_SIZE + 64)
        ctypes.cast(cinfo, ctypes.POINTER(ctypes.c_void_p))[0] = L.jpeg_std_error(err)
        L.jpeg_CreateCompress(cinfo, JPEG_LIB_VERSION, STRUCT_SIZE)  # exits with a message on a size mismatch
        out = ctypes.c_void_p(0)
        outsize = ctypes.c_ulong(0)
        L.jpeg_mem_dest(cinfo, ctypes.byref(out), ctypes.byref(outsize))
        struct.pack_into("IIii", cinfo, OFF_IMAGE_WIDTH, w, h, 3, JCS_RGB)
        L.jpeg_set_defaults(cinfo)
        # layout checks against what the library just wrote
        assert struct.unpack_from("d", cinfo, OFF_INPUT_GAMMA)[0] == 1.0
        assert struct.unpack_from("i", cinfo, OFF_DCT_METHOD)[0] == JDCT_ISLOW
        assert cinfo.raw[OFF_WRITE_JFIF:OFF_WRITE_JFIF + 12] == bytes([1, 0, 0, 0, 1, 1, 0, 0, 1, 0, 1, 0])
        comp = struct.unpack_from("Q", cinfo, OFF_COMP_INFO)[0]
        dflt = [tuple(ctypes.cast(comp + 96 * c + 8, ctypes.POINTER(ctypes.c_int))[0:2]) for c in range(3)]
        assert dflt == [(2, 2), (1, 1), (1, 1)], dflt  # jpeg_set_colorspace(YCbCr) default = 4:2:0 = TJSAMP_420
        if colorspace is not None:
            L.jpeg_set_colorspace(cinfo, colorspace)  # (resets the sampling factors to 1x1 for RGB)
        assert struct.unpack_from("i", cinfo, OFF_NUM_COMPONENTS)[0] == 3
        assert struct.unpack_from("Ii", cinfo, OFF_RESTART_INTERVAL) == (0, 0) and struct.unpack_from("i", cinfo, OFF_OPTIMIZE_CODING)[0] == 0
        assert struct.unpack_from("i", cinfo, OFF_WRITE_ADOBE)[0] == (1 if colorspace == JCS_RGB else 0)
        if samp is not None:
            for c, (sh, sv) in enumerate(samp):
                p = ctypes.cast(comp + 96 * c + 8, ctypes.POINTER(ctypes.c_int))
                p[0], p[1] = sh, sv
        L.jpeg_set_quality(cinfo, quality, 1)
        struct.pack_into("i", cinfo, OFF_DCT_METHOD, dct)
        struct.pack_into("Ii", cinfo, OFF_RESTART_INTERVAL, restart_mcus, restart_rows)
        if optimize:
            struct.pack_into("i", cinfo, OFF_OPTIMIZE_CODING, 1)
        if write_jfif is not None:
            struct.pack_into("i", cinfo, OFF_WRITE_JFIF, int(write_jfif))
        if write_adobe is not None:
            struct.pack_into("i", cinfo, OFF_WRITE_ADOBE, int(write_adobe))
        if progressive:
            L.jpeg_simple_progression(cinfo)
        if scans is not None:
            # a scan script (jpeg_scan_info: comps_in_scan, component_index[4], Ss, Se, Ah, Al), e.g. one sequential scan per
            # component: a NON-INTERLEAVED baseline file -- its scans run over ceil(width / 8) blocks of the component, not over MCUs
            assert struct.unpack_from("i", cinfo, OFF_NUM_SCANS)[0] == 0 and struct.unpack_from("Q", cinfo, OFF_NUM_SCANS + 8)[0] == 0
            arr = (ctypes.c_int * (9 * len(scans)))()
            for k, (comps, ss, se, ah, al) in enumerate(scans):
                arr[9 * k] = len(comps)
                for q, c in enumerate(comps):
                    arr[9 * k + 1 + q] = c
                arr[9 * k + 5:9 * k + 9] = [ss, se, ah, al]
            self._keep = arr
            struct.pack_into("i", cinfo, OFF_NUM_SCANS, len(scans))
            struct.pack_into("Q", cinfo, OFF_NUM_SCANS + 8, ctypes.addressof(arr))
        L.jpeg_start_compress(cinfo, 1)
        rows = (ctypes.c_void_p * h)(*[rgb.ctypes.data + y * 3 * w for y in range(h)])
        done = 0
        while done < h:
            done += L.jpeg_write_scanlines(cinfo, ctypes.byref(rows, done * ctypes.sizeof(ctypes.c_void_p)), h - done)
        L.jpeg_finish_compress(cinfo)
        data = ctypes.string_at(out.value, outsize.value)
        L.jpeg_destroy_compress(cinfo)
        return data


def pil_islow(rgb, quality=95):
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", quality=quality, subsampling="4:2:0", optimize=False)
    return b.getvalue()


def test_images():
    """Small frames: MCU-aligned, ragged right / bottom edges (dummy blocks, replicated samples), camera-like
    content, hard edges (green rectangle outlines as drawn by N1), saturated noise."""
    from infercam_onnx_amd import synth

    rng = np.random.default_rng(20260)
    imgs = {}
    frame = synth.synth_frame(synth.DEFAULT_FRAME_SEED, 2, 320, 240)
    imgs["cam_160x112"] = frame[40:152, 60:220].copy()
    imgs["cam_150x100"] = frame[30:130, 20:170].copy()       # Y: 19 x 13 blocks -> dummy column and dummy row
    imgs["cam_37x29"] = frame[100:129, 100:137].copy()       # odd width and height
    imgs["cam_64x52"] = frame[10:62, 200:264].copy()         # 7 block rows: a dummy row of Y blocks
    imgs["cam_40x24"] = frame[60:84, 30:70].copy()           # 5 block columns: a dummy column
    rect = frame[0:96, 0:128].copy()
    rect[10, 20:90] = rect[70, 20:90] = (0, 255, 0)
    rect[10:71, 20] = rect[10:71, 89] = (0, 255, 0)
    imgs["rect_128x96"] = rect
    imgs["noise_48x32"] = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    chk = np.zeros((32, 32, 3), np.uint8)
    chk[::2, 1::2] = 255
    chk[1::2, ::2] = 255
    imgs["checker_32x32"] = chk
    sat = np.zeros((24, 56, 3), np.uint8)
    sat[:, 28:] = 255
    sat[12:, :, 1] = 255 - sat[12:, :, 1]
    imgs["steps_56x24"] = sat
    imgs["flat_16x16"] = np.full((16, 16, 3), 200, np.uint8)
    imgs["one_1x1"] = np.array([[[12, 250, 7]]], np.uint8)
    return imgs


def main():
    enc = TurboEncoder()
    out = {}
    for name, rgb in test_images().items():
        fast = enc.encode(rgb, 95, JDCT_IFAST)
        slow = enc.encode(rgb, 95, JDCT_ISLOW)
        assert slow == pil_islow(rgb, 95), "libjpeg-turbo 2.1.2 (system) and 3.1.x (PIL) disagree on the ISLOW stream of " + name
        out[name + "/rgb"] = rgb
        out[name + "/ifast"] = np.frombuffer(fast, np.uint8)
        out[name + "/islow"] = np.frombuffer(slow, np.uint8)
        print("%-16s %4dx%-4d ifast %6d B  islow %6d B  %s" % (name, rgb.shape[1], rgb.shape[0], len(fast), len(slow),
                                                                  "(differ)" if fast != slow else "(same bytes)"))
    path = os.path.join(ROOT, "tests", "golden", "encode_q95_420.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
