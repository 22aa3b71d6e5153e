"""UltraFace-RFB topology table (what `model.run` computes, infer_server/src/nn.rs:181).

The network is defined by the ONNX file the reference downloads (nn.rs:21-22), exported from
Linzaer/Ultra-Light-Fast-Generic-Face-Detector-1MB `Mb_Tiny_RFB_fd`.
This table is the host-side description used to pack weights for the HIP kernels and to
generate synthetic weights / ONNX files; the C++ side carries the same table in
csrc/topology.hpp.
"""
from collections import namedtuple

ConvSpec = namedtuple("ConvSpec", "name cin cout k stride pad dil groups relu src")

# src: index of the conv whose (block) output feeds this one; -1 network input; -2 RFB concat
CONVS = [
    ConvSpec("m0.conv_bn", 3, 16, 3, 2, 1, 1, 1, 1, -1),
    ConvSpec("m1.dw", 16, 16, 3, 1, 1, 1, 16, 1, 0),
    ConvSpec("m1.pw", 16, 32, 1, 1, 0, 1, 1, 1, 1),
    ConvSpec("m2.dw", 32, 32, 3, 2, 1, 1, 32, 1, 2),
    ConvSpec("m2.pw", 32, 32, 1, 1, 0, 1, 1, 1, 3),
    ConvSpec("m3.dw", 32, 32, 3, 1, 1, 1, 32, 1, 4),
    ConvSpec("m3.pw", 32, 32, 1, 1, 0, 1, 1, 1, 5),
    ConvSpec("m4.dw", 32, 32, 3, 2, 1, 1, 32, 1, 6),
    ConvSpec("m4.pw", 32, 64, 1, 1, 0, 1, 1, 1, 7),
    ConvSpec("m5.dw", 64, 64, 3, 1, 1, 1, 64, 1, 8),
    ConvSpec("m5.pw", 64, 64, 1, 1, 0, 1, 1, 1, 9),
    ConvSpec("m6.dw", 64, 64, 3, 1, 1, 1, 64, 1, 10),
    ConvSpec("m6.pw", 64, 64, 1, 1, 0, 1, 1, 1, 11),
    ConvSpec("m7.rfb.b0.0", 64, 8, 1, 1, 0, 1, 1, 0, 12),
    ConvSpec("m7.rfb.b0.1", 8, 16, 3, 1, 1, 1, 1, 1, 13),
    ConvSpec("m7.rfb.b0.2", 16, 16, 3, 1, 2, 2, 1, 0, 14),
    ConvSpec("m7.rfb.b1.0", 64, 8, 1, 1, 0, 1, 1, 0, 12),
    ConvSpec("m7.rfb.b1.1", 8, 16, 3, 1, 1, 1, 1, 1, 16),
    ConvSpec("m7.rfb.b1.2", 16, 16, 3, 1, 3, 3, 1, 0, 17),
    ConvSpec("m7.rfb.b2.0", 64, 8, 1, 1, 0, 1, 1, 0, 12),
    ConvSpec("m7.rfb.b2.1", 8, 12, 3, 1, 1, 1, 1, 1, 19),
    ConvSpec("m7.rfb.b2.2", 12, 16, 3, 1, 1, 1, 1, 1, 20),
    ConvSpec("m7.rfb.b2.3", 16, 16, 3, 1, 5, 5, 1, 0, 21),
    ConvSpec("m7.rfb.ConvLinear", 48, 64, 1, 1, 0, 1, 1, 0, -2),
    ConvSpec("m7.rfb.shortcut", 64, 64, 1, 1, 0, 1, 1, 0, 12),  # + ConvLinear, relu
    ConvSpec("cls0.dw", 64, 64, 3, 1, 1, 1, 64, 1, 24),
    ConvSpec("cls0.pw", 64, 6, 1, 1, 0, 1, 1, 0, 25),
    ConvSpec("reg0.dw", 64, 64, 3, 1, 1, 1, 64, 1, 24),
    ConvSpec("reg0.pw", 64, 12, 1, 1, 0, 1, 1, 0, 27),
    ConvSpec("m8.dw", 64, 64, 3, 2, 1, 1, 64, 1, 24),
    ConvSpec("m8.pw", 64, 128, 1, 1, 0, 1, 1, 1, 29),
    ConvSpec("m9.dw", 128, 128, 3, 1, 1, 1, 128, 1, 30),
    ConvSpec("m9.pw", 128, 128, 1, 1, 0, 1, 1, 1, 31),
    ConvSpec("m10.dw", 128, 128, 3, 1, 1, 1, 128, 1, 32),
    ConvSpec("m10.pw", 128, 128, 1, 1, 0, 1, 1, 1, 33),
    ConvSpec("cls1.dw", 128, 128, 3, 1, 1, 1, 128, 1, 34),
    ConvSpec("cls1.pw", 128, 4, 1, 1, 0, 1, 1, 0, 35),
    ConvSpec("reg1.dw", 128, 128, 3, 1, 1, 1, 128, 1, 34),
    ConvSpec("reg1.pw", 128, 8, 1, 1, 0, 1, 1, 0, 37),
    ConvSpec("m11.dw", 128, 128, 3, 2, 1, 1, 128, 1, 34),
    ConvSpec("m11.pw", 128, 256, 1, 1, 0, 1, 1, 1, 39),
    ConvSpec("m12.dw", 256, 256, 3, 1, 1, 1, 256, 1, 40),
    ConvSpec("m12.pw", 256, 256, 1, 1, 0, 1, 1, 1, 41),
    ConvSpec("cls2.dw", 256, 256, 3, 1, 1, 1, 256, 1, 42),
    ConvSpec("cls2.pw", 256, 4, 1, 1, 0, 1, 1, 0, 43),
    ConvSpec("reg2.dw", 256, 256, 3, 1, 1, 1, 256, 1, 42),
    ConvSpec("reg2.pw", 256, 8, 1, 1, 0, 1, 1, 0, 45),
    ConvSpec("extra.0", 256, 64, 1, 1, 0, 1, 1, 1, 42),
    ConvSpec("extra.2.dw", 64, 64, 3, 2, 1, 1, 64, 1, 47),
    ConvSpec("extra.2.pw", 64, 256, 1, 1, 0, 1, 1, 1, 48),
    ConvSpec("cls3", 256, 6, 3, 1, 1, 1, 1, 0, 49),
    ConvSpec("reg3", 256, 12, 3, 1, 1, 1, 1, 0, 49),
]

NUM_CONV = len(CONVS)
STRIDES = (8, 16, 32, 64)
NUM_ANCHORS = (3, 2, 2, 3)
MIN_BOXES = ((10, 16, 24), (32, 48), (64, 96), (128, 192, 256))
CLS_LAYERS = (26, 36, 44, 50)
REG_LAYERS = (28, 38, 46, 51)
CENTER_VARIANCE = 0.1
SIZE_VARIANCE = 0.2

# UltrafaceVariant::width_height (nn.rs:36-41)
VARIANTS = {640: (640, 480), 320: (320, 240)}


def weight_count(spec):
    return spec.cout * (spec.cin // spec.groups) * spec.k * spec.k


def total_weight_floats():
    """Packed blob: for each conv w[cout][cin/g][k][k] then b[cout]."""
    return sum(weight_count(s) + s.cout for s in CONVS)


def weight_offsets():
    """[(w_off, b_off)] float offsets into the packed blob."""
    offs, o = [], 0
    for s in CONVS:
        offs.append((o, o + weight_count(s)))
        o += weight_count(s) + s.cout
    return offs


def conv_out(n, s):
    return (n + 2 * s.pad - s.dil * (s.k - 1) - 1) // s.stride + 1


def layer_hw(width, height):
    """[(in_h, in_w, out_h, out_w)] per conv for a model input size."""
    out = []
    for s in CONVS:
        if s.src == -1:
            ih, iw = height, width
        elif s.src == -2:
            ih, iw = out[15][2], out[15][3]
        else:
            ih, iw = out[s.src][2], out[s.src][3]
        out.append((ih, iw, conv_out(ih, s), conv_out(iw, s)))
    return out


def feature_maps(width, height):
    return [(-(-width // st), -(-height // st)) for st in STRIDES]


def num_priors(width, height):
    return sum(fw * fh * a for (fw, fh), a in zip(feature_maps(width, height), NUM_ANCHORS))


def macs(width, height):
    hw = layer_hw(width, height)
    return sum(oh * ow * s.cout * (s.cin // s.groups) * s.k * s.k for s, (_, _, oh, ow) in zip(CONVS, hw))
