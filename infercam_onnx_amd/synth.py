"""Synthetic inputs for the hot path: seeded weights, priors and camera-like JPEG frames.

Neither box has the real `version-RFB-{320,640}.onnx` (the reference downloads it at run
time, infer_server/src/nn.rs:21-22,155-162) and there is no camera (cam_sender), so tests and
the benchmark use the generators here (SURVEY.md section 8d "synthetic inputs" / "weights").
"""
import io

import numpy as np

from . import topology as T

DEFAULT_WEIGHT_SEED = 1234
DEFAULT_FRAME_SEED = 0x5EED0000

# Shift applied to the face-class (column 1) bias of each cls head so that, with the seeded
# He-normal weights and the synthetic frames below, roughly 0.5 % of the 17 640 priors exceed
# confidence 0.5 -- keeps threshold/sort/NMS realistically loaded.  Calibrated once with
# tools/calibrate_cls_shift.py against the CPU oracle at 640x480 (values are part of the
# synthetic-weight definition: changing them changes the golden fixtures).
CLS_BIAS_SHIFT = (-8.3, -4.1, -4.5, -5.7)


def gen_priors(width, height):
    """Upstream generate_priors (float64 -> f32, clamped); `[K,4]` (cx, cy, w, h)."""
    out = []
    for (fw, fh), boxes in zip(T.feature_maps(width, height), T.MIN_BOXES):
        shrink_w, shrink_h = width / fw, height / fh
        scale_w, scale_h = width / shrink_w, height / shrink_h
        for j in range(fh):
            for i in range(fw):
                xc, yc = (i + 0.5) / scale_w, (j + 0.5) / scale_h
                for m in boxes:
                    out.append((xc, yc, m / width, m / height))
    return np.clip(np.asarray(out, np.float64).astype(np.float32), 0.0, 1.0)


def synthetic_weights(seed=DEFAULT_WEIGHT_SEED, cls_bias_shift=CLS_BIAS_SHIFT):
    """Packed f32 blob (`topology.total_weight_floats()` floats): He-normal N(0, 2/fan_in)
    weights, N(0, 0.01^2) biases, one numpy Generator per layer seeded `seed + layer`."""
    blob = np.empty(T.total_weight_floats(), np.float32)
    for i, (s, (wo, bo)) in enumerate(zip(T.CONVS, T.weight_offsets())):
        rng = np.random.default_rng(seed + i)
        fan_in = (s.cin // s.groups) * s.k * s.k
        n = T.weight_count(s)
        blob[wo:wo + n] = (rng.standard_normal(n) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        b = (rng.standard_normal(s.cout) * 0.01).astype(np.float32)
        if i in T.CLS_LAYERS and cls_bias_shift is not None:
            b[1::2] += np.float32(cls_bias_shift[T.CLS_LAYERS.index(i)])
        blob[bo:bo + s.cout] = b
    return blob


def bn_folded_like_weights(seed=7, spread=4.0, bias_sigma=0.5):
    """Weights with the dynamic range of a BatchNorm-folded checkpoint (w' = w * gamma / sigma per output channel,
    b' = beta - mu * gamma / sigma) instead of the uniform He-normal scale of `synthetic_weights`: every output channel
    gets its own factor, log-uniform over a ratio of spread^2 and normalised to unit mean square per layer (activations
    neither explode nor die over the 13 blocks), and biases are N(0, bias_sigma^2) -- tens of times the He blob's.  The real zoo file is not
    available offline (nn.rs:21-22 downloads it); this is the closest the parity tests can get to its value ranges."""
    blob = np.empty(T.total_weight_floats(), np.float32)
    for i, (s, (wo, bo)) in enumerate(zip(T.CONVS, T.weight_offsets())):
        rng = np.random.default_rng([int(seed), i])
        fan_in = (s.cin // s.groups) * s.k * s.k
        n = T.weight_count(s)
        w = (rng.standard_normal(n) * np.sqrt(2.0 / fan_in)).reshape(s.cout, -1)
        scale = np.exp(rng.uniform(-np.log(spread), np.log(spread), s.cout))
        scale /= np.sqrt(np.mean(scale * scale))  # (the layer keeps the He blob's mean output power: nothing explodes)
        is_head = i in T.CLS_LAYERS or i in T.REG_LAYERS
        if is_head:
            scale[:] = 1.0  # (the heads feed softmax / exp: keep their logits in a trained model's range)
        blob[wo:wo + n] = (w * scale[:, None]).astype(np.float32).ravel()
        blob[bo:bo + s.cout] = (rng.standard_normal(s.cout) * (0.05 if is_head else bias_sigma)).astype(np.float32)
    return blob


def layer_params(blob, i):
    """(w [cout, cin/g, k, k], b [cout]) views into a packed blob."""
    s = T.CONVS[i]
    wo, bo = T.weight_offsets()[i]
    n = T.weight_count(s)
    return blob[wo:wo + n].reshape(s.cout, s.cin // s.groups, s.k, s.k), blob[bo:bo + s.cout]


def synth_frame(seed, idx, width, height):
    """Camera-like RGB frame: smooth gradient + 0-6 skin-tone ellipses with darker eye / mouth
    ellipses + Gaussian noise (sigma 3).  Deterministic in (seed, idx, width, height)."""
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, int(idx), int(width), int(height)])
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    u, v = xx / max(width - 1, 1), yy / max(height - 1, 1)
    c0 = rng.uniform(40, 200, 3).astype(np.float32)
    gx = rng.uniform(-60, 60, 3).astype(np.float32)
    gy = rng.uniform(-60, 60, 3).astype(np.float32)
    img = c0[None, None, :] + u[..., None] * gx[None, None, :] + v[..., None] * gy[None, None, :]

    def ellipse(cx, cy, rx, ry, colour, soft):
        d = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2
        a = np.clip((1.0 - d) / soft, 0.0, 1.0)[..., None]
        return img * (1 - a) + np.asarray(colour, np.float32)[None, None, :] * a

    for _ in range(int(rng.integers(0, 7))):
        s = float(rng.uniform(0.04, 0.22)) * height
        cx, cy = float(rng.uniform(0.1, 0.9)) * width, float(rng.uniform(0.15, 0.85)) * height
        skin = np.array([224, 172, 140], np.float32) * float(rng.uniform(0.6, 1.1))
        img = ellipse(cx, cy, 0.75 * s, s, skin, 0.25)
        dark = skin * 0.35
        img = ellipse(cx - 0.3 * s, cy - 0.25 * s, 0.14 * s, 0.09 * s, dark, 0.5)
        img = ellipse(cx + 0.3 * s, cy - 0.25 * s, 0.14 * s, 0.09 * s, dark, 0.5)
        img = ellipse(cx, cy + 0.45 * s, 0.3 * s, 0.1 * s, dark * 1.3, 0.5)
    img = img + rng.standard_normal(img.shape).astype(np.float32) * 3.0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def encode_jpeg(rgb, quality=90, subsampling="4:2:0", restart_rows=0, progressive=False, optimize=False):
    """Baseline-Huffman JPEG with the standard Annex-K tables (what a UVC MJPG camera sends),
    via the libjpeg-turbo bundled with PIL.  `restart_rows` > 0 adds a DRI of that many MCU rows."""
    from PIL import Image, ImageFile

    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 1 << 25)
    kw = {}
    if restart_rows:
        kw["restart_marker_rows"] = int(restart_rows)
    if progressive:
        kw["progressive"] = True
    if optimize:
        kw["optimize"] = True
    bio = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(rgb)).save(bio, "JPEG", quality=quality, subsampling=subsampling, **kw)
    return bio.getvalue()


def strip_dht(jpeg):
    """The same stream without its DHT segments: what a UVC camera's MJPG frame looks like (the tables
    are then the Annex-K defaults, which PIL / libjpeg-turbo also wrote into a non-optimised stream)."""
    out, i = bytearray(jpeg[:2]), 2
    while i + 4 <= len(jpeg):
        if jpeg[i] != 0xFF:
            raise ValueError("not at a marker")
        marker = jpeg[i + 1]
        if marker == 0xDA:  # SOS: entropy-coded data follows, copy the rest
            out += jpeg[i:]
            return bytes(out)
        seg = 2 + ((jpeg[i + 2] << 8) | jpeg[i + 3])
        if marker != 0xC4:
            out += jpeg[i:i + seg]
        i += seg
    raise ValueError("no SOS marker")


def synth_jpeg_pool(stream_id, count, width, height, quality=90, subsampling="4:2:0", restart_rows=0):
    """Pool of `count` distinct frames for one camera stream (seed 0x5EED0000 + stream_id)."""
    seed = DEFAULT_FRAME_SEED + int(stream_id)
    return [encode_jpeg(synth_frame(seed, i, width, height), quality, subsampling, restart_rows) for i in range(count)]


# ---------------------------------------------------------------------------------------------
# Minimal ONNX writer (protobuf wire format by hand; onnx.proto3 field numbers) so the loader
# (csrc/onnx_loader.cpp, replacing tract's ONNX front end, nn.rs:166-172) can be exercised
# without the real model file.
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _vi(field, v):
    return _varint(field << 3) + _varint(v)


def _tensor(name, arr):
    arr = np.ascontiguousarray(arr)
    dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 7}[arr.dtype]
    return b"".join(_vi(1, d) for d in arr.shape) + _vi(2, dt) + _ld(8, name.encode()) + _ld(9, arr.tobytes())


def _attr_ints(name, vals):
    return _ld(1, name.encode()) + b"".join(_vi(8, v) for v in vals) + _vi(20, 7)


def _attr_int(name, v):
    return _ld(1, name.encode()) + _vi(3, v) + _vi(20, 2)


def _attr_float(name, v):
    return _ld(1, name.encode()) + _varint((2 << 3) | 5) + np.float32(v).tobytes() + _vi(20, 1)


def _attr_tensor(name, t):
    return _ld(1, name.encode()) + _ld(5, t) + _vi(20, 4)


def _node(op, ins, outs, attrs=()):
    return (b"".join(_ld(1, i.encode()) for i in ins) + b"".join(_ld(2, o.encode()) for o in outs) + _ld(4, op.encode()) +
            b"".join(_ld(5, a) for a in attrs))


def write_onnx(path, blob, width, height, with_batchnorm=False, seed=99, priors_as="constant"):
    """Writes an UltraFace-RFB-shaped ONNX file holding `blob`'s convolutions.  With
    `with_batchnorm`, convs 0..24 (the BasicConv / conv_bn / conv_dw blocks upstream) are emitted
    as bias-free Conv + BatchNormalization whose folding reproduces `blob`; returns the blob the
    loader is expected to produce."""
    rng = np.random.default_rng(seed)
    nodes, inits = [], []
    expect = np.array(blob, np.float32, copy=True)
    for i, (s, (wo, bo)) in enumerate(zip(T.CONVS, T.weight_offsets())):
        w, b = layer_params(blob, i)
        attrs = [_attr_ints("dilations", [s.dil] * 2), _attr_int("group", s.groups),
                 _attr_ints("kernel_shape", [s.k] * 2), _attr_ints("pads", [s.pad] * 4),
                 _attr_ints("strides", [s.stride] * 2)]
        x, wn, bn, y = "x%d" % i, "w%d" % i, "b%d" % i, "y%d" % i
        if with_batchnorm and i <= 24:
            gamma = rng.uniform(0.5, 1.5, s.cout).astype(np.float32)
            beta = rng.normal(0, 0.1, s.cout).astype(np.float32)
            mean = rng.normal(0, 0.1, s.cout).astype(np.float32)
            var = rng.uniform(0.5, 1.5, s.cout).astype(np.float32)
            eps = np.float32(1e-5)
            inits += [_tensor(wn, w)] + [_tensor("%s_%s" % (bn, k), v) for k, v in
                                         (("g", gamma), ("b", beta), ("m", mean), ("v", var))]
            nodes.append(_node("Conv", [x, wn], [y + "_c"], attrs))
            nodes.append(_node("BatchNormalization", [y + "_c", bn + "_g", bn + "_b", bn + "_m", bn + "_v"], [y],
                               [_attr_float("epsilon", eps), _attr_float("momentum", 0.9)]))
            sc = gamma / np.sqrt(var + eps)
            n = T.weight_count(s)
            expect[wo:wo + n] = (w * sc[:, None, None, None]).ravel()
            expect[bo:bo + s.cout] = beta + (np.float32(0) - mean) * sc
        else:
            inits += [_tensor(wn, w), _tensor(bn, b)]
            nodes.append(_node("Conv", [x, wn, bn], [y], attrs))
        if s.relu:
            nodes.append(_node("Relu", [y], [y + "_r"]))
    pri = gen_priors(width, height)[None]
    if priors_as == "constant":
        nodes.append(_node("Constant", [], ["priors"], [_attr_tensor("value", _tensor("", pri))]))
    elif priors_as == "initializer":
        inits.append(_tensor("priors", pri))
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"ultraface-rfb-synthetic") + b"".join(_ld(5, t) for t in inits)
    model = _vi(1, 4) + _ld(2, b"infercam_onnx_amd.synth") + _ld(7, graph) + _ld(8, _vi(2, 9))
    with open(path, "wb") as f:
        f.write(model)
    return expect
