"""ctypes binding of oracle/libufd_oracle.so (plain-C restatement of the reference path).

TEST INFRASTRUCTURE ONLY -- see oracle/ufd_oracle.h for what each function restates
(reference file:line) and what pins it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libufd_oracle.so")

NUM_CONV = 52


class JpegInfo(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("ncomp", ctypes.c_int),
                ("progressive", ctypes.c_int), ("hsamp", ctypes.c_int * 4), ("vsamp", ctypes.c_int * 4),
                ("restart_interval", ctypes.c_int)]


class ConvSpec(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("cin", "cout", "k", "stride", "pad", "dil", "groups", "relu")]


class Det(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("x_tl", "y_tl", "x_br", "y_br", "conf")]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        cmd = ["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + ["libufd_oracle.so"]
        subprocess.check_call(cmd)
    return _LIB_PATH


sz_t = ctypes.c_size_t
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        c_int, c_f, vp, sz = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
        L.ufo_jpeg_probe.argtypes = [vp, sz, ctypes.POINTER(JpegInfo)]
        L.ufo_jpeg_decode_rgb.argtypes = [vp, sz, vp, c_int, c_int]
        L.ufo_resize_triangle_rgb.argtypes = [vp, c_int, c_int, vp, c_int, c_int]
        L.ufo_axis_taps.argtypes = [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), vp, c_int]
        L.ufo_normalize_nchw.argtypes = [vp, c_int, c_int, vp]
        L.ufo_normalize_nchw.restype = None
        L.ufo_conv_specs.restype = ctypes.POINTER(ConvSpec)
        L.ufo_num_priors.argtypes = [c_int, c_int]
        L.ufo_gen_priors.argtypes = [c_int, c_int, vp]
        L.ufo_gen_priors.restype = None
        L.ufo_weight_floats.restype = sz
        L.ufo_ultraface_forward.argtypes = [vp, c_int, c_int, vp, vp, vp, vp]
        L.ufo_ultraface_forward_layers.argtypes = [vp, c_int, c_int, vp, vp, vp, vp, ctypes.POINTER(vp)]
        L.ufo_layer_out_hw.argtypes = [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
        L.ufo_layer_out_hw.restype = None
        L.ufo_bbox_area.argtypes = [vp]
        L.ufo_bbox_area.restype = c_f
        L.ufo_iou.argtypes = [vp, vp]
        L.ufo_iou.restype = c_f
        L.ufo_postproc.argtypes = [vp, vp, c_int, c_f, c_f, vp, c_int]
        L.ufo_infer_rgb.argtypes = [vp, c_int, c_int, c_int, c_int, vp, vp, c_f, c_f, vp, c_int]
        L.ufo_infer_jpeg.argtypes = [vp, sz, c_int, c_int, vp, vp, c_f, c_f, vp, c_int]
        L.ufo_infer_jpeg_mt.argtypes = [vp, vp, c_int, c_int, ctypes.c_double, c_int, c_int, c_int, vp, vp, c_f, c_f, c_int, vp]
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.ufo_rect_of_det.argtypes = [vp, c_f, c_f, i64p, i64p, i64p, i64p]
        L.ufo_draw_hollow_rects.argtypes = [vp, c_int, c_int, vp, c_int, c_f, c_f]
        L.ufo_draw_hollow_rects.restype = None
        L.ufo_label_chars.argtypes = [c_f, vp]
        L.ufo_draw_labels.argtypes = [vp, c_int, c_int, vp, c_int, c_f, c_f]
        L.ufo_draw_labels.restype = None
        L.ufo_jpeg_quant_table.argtypes = [c_int, c_int, vp]
        L.ufo_jpeg_quant_table.restype = None
        L.ufo_jpeg_encode_bound.argtypes = [c_int, c_int]
        L.ufo_jpeg_encode_bound.restype = sz
        L.ufo_jpeg_encode_rgb.argtypes = [vp, c_int, c_int, c_int, c_int, vp, sz, ctypes.POINTER(sz)]
        L.ufo_jpeg_encode_coefficients.argtypes = [vp, c_int, c_int, c_int, c_int, vp]
        L.ufo_stream_item.argtypes = [vp, sz, vp, sz]
        L.ufo_stream_item.restype = sz
        L.ufo_annotate_encode_jpeg.argtypes = [vp, sz, c_int, c_int, vp, vp, c_f, c_f, c_f, c_f, c_int, vp, c_int, vp, sz,
                                               ctypes.POINTER(sz)]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, what, rc):
        super().__init__("%s failed: rc=%d" % (what, rc))
        self.rc = rc


def _chk(rc, what):
    if rc < 0:
        raise OracleError(what, rc)
    return rc


def _buf(b):
    return (ctypes.c_char * max(len(b), 1)).from_buffer_copy(bytes(b) if len(b) else b"\0")


def jpeg_probe(data):
    info = JpegInfo()
    _chk(lib().ufo_jpeg_probe(_buf(data), len(data), ctypes.byref(info)), "jpeg_probe")
    return info


def jpeg_decode_rgb(data):
    """A1: JPEG bytes -> HxWx3 uint8 (turbojpeg::decompress_image semantics)."""
    info = jpeg_probe(data)
    out = np.empty((info.height, info.width, 3), np.uint8)
    _chk(lib().ufo_jpeg_decode_rgb(_buf(data), len(data), out.ctypes.data, info.width, info.height), "jpeg_decode")
    return out


def resize_triangle(rgb, dw, dh):
    """A2/A3: image::imageops::resize(.., Triangle)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    sh, sw, _ = rgb.shape
    out = np.empty((dh, dw, 3), np.uint8)
    _chk(lib().ufo_resize_triangle_rgb(rgb.ctypes.data, sw, sh, out.ctypes.data, dw, dh), "resize")
    return out


def axis_taps(S, D, o):
    """A3: (first source index, normalised f32 weights) of output index o along an axis of S -> D samples."""
    cap = int(np.ceil(2.0 * max(S / D, 1.0))) + 8
    w = np.zeros(cap, np.float32)
    left, n = ctypes.c_int(), ctypes.c_int()
    _chk(lib().ufo_axis_taps(S, D, o, ctypes.byref(left), ctypes.byref(n), w.ctypes.data, cap), "axis_taps")
    return left.value, w[:n.value].copy()


def normalize_nchw(rgb):
    """A4: HWC u8 -> [3,H,W] f32."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.empty((3, h, w), np.float32)
    lib().ufo_normalize_nchw(rgb.ctypes.data, w, h, out.ctypes.data)
    return out


def conv_specs():
    p = lib().ufo_conv_specs()
    return [dict((n, getattr(p[i], n)) for n, _ in ConvSpec._fields_) for i in range(NUM_CONV)]


def num_priors(w, h):
    return lib().ufo_num_priors(w, h)


def gen_priors(w, h):
    out = np.empty((num_priors(w, h), 4), np.float32)
    lib().ufo_gen_priors(w, h, out.ctypes.data)
    return out


def weight_floats():
    return int(lib().ufo_weight_floats())


def layer_out_hw(layer, w, h):
    oh, ow = ctypes.c_int(), ctypes.c_int()
    lib().ufo_layer_out_hw(layer, w, h, ctypes.byref(oh), ctypes.byref(ow))
    return oh.value, ow.value


def forward(inp, weights, priors, layers=False):
    """A6: [3,H,W] f32 -> (scores [K,2], boxes [K,4]) (+ list of per-conv outputs)."""
    inp = np.ascontiguousarray(inp, np.float32)
    weights = np.ascontiguousarray(weights, np.float32)
    priors = np.ascontiguousarray(priors, np.float32)
    _, h, w = inp.shape
    K = num_priors(w, h)
    assert priors.shape == (K, 4) and weights.size == weight_floats()
    scores = np.empty((K, 2), np.float32)
    boxes = np.empty((K, 4), np.float32)
    if not layers:
        _chk(lib().ufo_ultraface_forward(inp.ctypes.data, w, h, weights.ctypes.data, priors.ctypes.data,
                                         scores.ctypes.data, boxes.ctypes.data), "forward")
        return scores, boxes
    specs = conv_specs()
    outs = []
    for i, s in enumerate(specs):
        oh, ow = layer_out_hw(i, w, h)
        outs.append(np.empty((s["cout"], oh, ow), np.float32))
    ptrs = (ctypes.c_void_p * NUM_CONV)(*[o.ctypes.data for o in outs])
    _chk(lib().ufo_ultraface_forward_layers(inp.ctypes.data, w, h, weights.ctypes.data, priors.ctypes.data,
                                            scores.ctypes.data, boxes.ctypes.data, ptrs), "forward_layers")
    return scores, boxes, outs


def bbox_area(b):
    b = np.ascontiguousarray(b, np.float32)
    return float(lib().ufo_bbox_area(b.ctypes.data))


def iou(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return float(lib().ufo_iou(a.ctypes.data, b.ctypes.data))


def _dets(arr, n):
    return np.array([[d.x_tl, d.y_tl, d.x_br, d.y_br, d.conf] for d in arr[:n]], np.float32).reshape(-1, 5)


def postproc(scores, boxes, min_confidence=0.5, max_iou=0.5):
    """A7-A10: returns [n,5] (x_tl,y_tl,x_br,y_br,conf), descending confidence."""
    scores = np.ascontiguousarray(scores, np.float32)
    boxes = np.ascontiguousarray(boxes, np.float32)
    K = scores.shape[0]
    out = (Det * max(K, 1))()
    n = _chk(lib().ufo_postproc(scores.ctypes.data, boxes.ctypes.data, K, min_confidence, max_iou, out, K), "postproc")
    return _dets(out, n)


def infer_rgb(rgb, model_w, model_h, weights, priors, min_confidence=0.5, max_iou=0.5):
    """InferModel::run (nn.rs:178-186)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    weights = np.ascontiguousarray(weights, np.float32)
    priors = np.ascontiguousarray(priors, np.float32)
    h, w, _ = rgb.shape
    K = priors.shape[0]
    out = (Det * K)()
    n = _chk(lib().ufo_infer_rgb(rgb.ctypes.data, w, h, model_w, model_h, weights.ctypes.data, priors.ctypes.data,
                                 min_confidence, max_iou, out, K), "infer_rgb")
    return _dets(out, n)


def infer_jpeg(data, model_w, model_h, weights, priors, min_confidence=0.5, max_iou=0.5):
    """Inferer::run decode -> infer (inferer.rs:35-37)."""
    weights = np.ascontiguousarray(weights, np.float32)
    priors = np.ascontiguousarray(priors, np.float32)
    K = priors.shape[0]
    out = (Det * K)()
    n = _chk(lib().ufo_infer_jpeg(_buf(data), len(data), model_w, model_h, weights.ctypes.data, priors.ctypes.data,
                                  min_confidence, max_iou, out, K), "infer_jpeg")
    return _dets(out, n)


def infer_jpeg_many_threads(jpegs, total, threads, model_w, model_h, weights, priors, min_confidence=0.5, max_iou=0.5,
                            budget_s=0.0):
    """bench.py's all-cores cpu_baseline leg: up to `total` frames (cycling over `jpegs`) on `threads` host
    threads inside the C library, no frame started after `budget_s` seconds; returns (frames done, detections found)."""
    weights = np.ascontiguousarray(weights, np.float32)
    priors = np.ascontiguousarray(priors, np.float32)
    bufs = [_buf(j) for j in jpegs]
    ptrs = (ctypes.c_void_p * len(bufs))(*[ctypes.addressof(b) for b in bufs])
    lens = (ctypes.c_size_t * len(bufs))(*[len(j) for j in jpegs])
    dets = ctypes.c_long()
    n = _chk(lib().ufo_infer_jpeg_mt(ptrs, lens, len(bufs), int(total), float(budget_s), int(threads), model_w, model_h, weights.ctypes.data,
                                     priors.ctypes.data, min_confidence, max_iou, priors.shape[0], ctypes.byref(dets)),
             "infer_jpeg_mt")
    return n, dets.value


# ---- N1: rectangles + JPEG re-encode (inferer.rs:38-40), multipart framing (lib.rs:48-57)
def _det_array(dets):
    dets = np.asarray(dets, np.float32).reshape(-1, 5)
    arr = (Det * max(len(dets), 1))()
    for i, d in enumerate(dets):
        arr[i].x_tl, arr[i].y_tl, arr[i].x_br, arr[i].y_br, arr[i].conf = (float(v) for v in d)
    return arr, len(dets)


def rect_of_det(det, label_w, label_h):
    """inferer.rs:66-76: inclusive pixel rectangle (left, top, right, bottom) of one detection, or None when
    Rect::of_size would assert."""
    arr, _ = _det_array([det])
    v = [ctypes.c_int64() for _ in range(4)]
    ok = lib().ufo_rect_of_det(arr, label_w, label_h, *[ctypes.byref(x) for x in v])
    return tuple(x.value for x in v) if ok else None


def draw_hollow_rects(rgb, dets, label_w, label_h):
    """draw_bboxes_on_image without the text; returns a new HxWx3 array."""
    out = np.ascontiguousarray(rgb, np.uint8).copy()
    arr, n = _det_array(dets)
    lib().ufo_draw_hollow_rects(out.ctypes.data, out.shape[1], out.shape[0], arr, n, label_w, label_h)
    return out


def label_text(confidence):
    """format!("{:.2}%", confidence * 100.0) (inferer.rs:88)."""
    buf = (ctypes.c_uint8 * 8)()
    n = lib().ufo_label_chars(float(confidence), buf)
    return "".join("0123456789.%"[buf[i]] for i in range(n))


def draw_labels(rgb, dets, label_w, label_h):
    """draw_bboxes_on_image: rectangle + confidence label per detection, in order; returns a new HxWx3 array."""
    out = np.ascontiguousarray(rgb, np.uint8).copy()
    arr, n = _det_array(dets)
    lib().ufo_draw_labels(out.ctypes.data, out.shape[1], out.shape[0], arr, n, label_w, label_h)
    return out


def quant_table(quality, chroma):
    out = np.empty(64, np.uint8)
    lib().ufo_jpeg_quant_table(quality, int(chroma), out.ctypes.data)
    return out


def jpeg_encode_rgb(rgb, quality=95, dct=-1):
    """turbojpeg::compress_image(&frame, quality, Sub2x2): dct 0 ISLOW, 1 IFAST, -1 tjCompress2's choice."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    cap = lib().ufo_jpeg_encode_bound(w, h)
    out = np.empty(cap, np.uint8)
    n = sz_t()
    _chk(lib().ufo_jpeg_encode_rgb(rgb.ctypes.data, w, h, quality, dct, out.ctypes.data, cap, ctypes.byref(n)), "jpeg_encode")
    return out[:n.value].tobytes()


def jpeg_encode_coefficients(rgb, quality=95, dct=-1):
    """Quantised coefficients [mcus][6][64] (natural order) of that stream."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    if dct < 0:
        dct = 0 if quality >= 96 else 1
    out = np.empty((((w + 15) // 16) * ((h + 15) // 16), 6, 64), np.int16)
    _chk(lib().ufo_jpeg_encode_coefficients(rgb.ctypes.data, w, h, quality, dct, out.ctypes.data), "jpeg_encode_coefficients")
    return out


def stream_item(jpeg):
    n = lib().ufo_stream_item(_buf(jpeg), len(jpeg), None, 0)
    out = ctypes.create_string_buffer(n)
    lib().ufo_stream_item(_buf(jpeg), len(jpeg), out, n)
    return out.raw


def annotate_encode_jpeg(data, model_w, model_h, weights, priors, label_w, label_h, min_confidence=0.5, max_iou=0.5, quality=95):
    """Inferer::run, inferer.rs:35-40: returns (detections [n,5], annotated JPEG bytes)."""
    weights = np.ascontiguousarray(weights, np.float32)
    priors = np.ascontiguousarray(priors, np.float32)
    info = jpeg_probe(data)
    K = priors.shape[0]
    dets = (Det * K)()
    cap = lib().ufo_jpeg_encode_bound(info.width, info.height)
    out = np.empty(cap, np.uint8)
    n_out = sz_t()
    n = _chk(lib().ufo_annotate_encode_jpeg(_buf(data), len(data), model_w, model_h, weights.ctypes.data, priors.ctypes.data,
                                            min_confidence, max_iou, label_w, label_h, quality, dets, K, out.ctypes.data, cap,
                                            ctypes.byref(n_out)), "annotate_encode_jpeg")
    return _dets(dets, n), out[:n_out.value].tobytes()
