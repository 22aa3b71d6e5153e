"""UltraFace "RFB" (Mb_Tiny_RFB_fd + SSD head) as a torch.nn.Module, exported to ONNX by torch's own
(legacy TorchScript) exporter -- TEST INFRASTRUCTURE.

Purpose: the ONNX loader (csrc/onnx_loader.cpp, replacing tract's ONNX front end,
infer_server/src/nn.rs:143-175) and the kernels' topology were only ever fed files written by the
repo's own protobuf writer (synth.write_onnx).  This module is the network the onnx/models zoo file
`version-RFB-{320,640}.onnx` was exported from (Linzaer/Ultra-Light-Fast-Generic-Face-Detector-1MB,
`vision/nn/mb_tiny_RFB.py`, `vision/ssd/ssd.py`, `vision/ssd/mb_tiny_RFB_fd.py`; the reference README
links it, README.md:139-145) written with upstream's module structure: BatchNorm un-folded, the softmax /
prior-decode tail inside the graph, priors embedded as a constant.  The file that comes out of
torch.onnx.export has the exporter's node order, names, Constant/initializer placement and BN
layout -- none of it chosen by this repo -- and the loader must reproduce the folded blob and the
priors from it, and the GPU must reproduce torch's own forward.

The real zoo weights are not available offline (nn.rs:21-22 downloads them): weights are seeded.
"""
import io
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- vision/nn/mb_tiny_RFB.py
class BasicConv(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, dilation=1, groups=1, relu=True, bn=True):
        super().__init__()
        self.out_channels = out_planes
        if bn:
            self.conv = nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding,
                                  dilation=dilation, groups=groups, bias=False)
            self.bn = nn.BatchNorm2d(out_planes, eps=1e-5, momentum=0.01, affine=True)
            self.relu = nn.ReLU(inplace=True) if relu else None
        else:
            self.conv = nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding,
                                  dilation=dilation, groups=groups, bias=True)
            self.bn = None
            self.relu = nn.ReLU(inplace=True) if relu else None

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        if self.relu is not None:
            x = self.relu(x)
        return x


class BasicRFB(nn.Module):
    def __init__(self, in_planes, out_planes, stride=1, scale=0.1, map_reduce=8, vision=1, groups=1):
        super().__init__()
        self.scale = scale
        self.out_channels = out_planes
        inter_planes = in_planes // map_reduce
        self.branch0 = nn.Sequential(
            BasicConv(in_planes, inter_planes, kernel_size=1, stride=1, groups=groups, relu=False),
            BasicConv(inter_planes, 2 * inter_planes, kernel_size=(3, 3), stride=stride, padding=(1, 1), groups=groups),
            BasicConv(2 * inter_planes, 2 * inter_planes, kernel_size=3, stride=1, padding=vision + 1, dilation=vision + 1,
                      relu=False, groups=groups))
        self.branch1 = nn.Sequential(
            BasicConv(in_planes, inter_planes, kernel_size=1, stride=1, groups=groups, relu=False),
            BasicConv(inter_planes, 2 * inter_planes, kernel_size=(3, 3), stride=stride, padding=(1, 1), groups=groups),
            BasicConv(2 * inter_planes, 2 * inter_planes, kernel_size=3, stride=1, padding=vision + 2, dilation=vision + 2,
                      relu=False, groups=groups))
        self.branch2 = nn.Sequential(
            BasicConv(in_planes, inter_planes, kernel_size=1, stride=1, groups=groups, relu=False),
            BasicConv(inter_planes, (inter_planes // 2) * 3, kernel_size=3, stride=1, padding=1, groups=groups),
            BasicConv((inter_planes // 2) * 3, 2 * inter_planes, kernel_size=3, stride=stride, padding=1, groups=groups),
            BasicConv(2 * inter_planes, 2 * inter_planes, kernel_size=3, stride=1, padding=vision + 4, dilation=vision + 4,
                      relu=False, groups=groups))
        self.ConvLinear = BasicConv(6 * inter_planes, out_planes, kernel_size=1, stride=1, relu=False)
        self.shortcut = BasicConv(in_planes, out_planes, kernel_size=1, stride=stride, relu=False)
        self.relu = nn.ReLU(inplace=False)

    def forward(self, x):
        x0 = self.branch0(x)
        x1 = self.branch1(x)
        x2 = self.branch2(x)
        out = torch.cat((x0, x1, x2), 1)
        out = self.ConvLinear(out)
        short = self.shortcut(x)
        out = out * self.scale + short
        out = self.relu(out)
        return out


class Mb_Tiny_RFB(nn.Module):
    def __init__(self, num_classes=2):
        super().__init__()
        self.base_channel = 8 * 2

        def conv_bn(inp, oup, stride):
            return nn.Sequential(nn.Conv2d(inp, oup, 3, stride, 1, bias=False), nn.BatchNorm2d(oup), nn.ReLU(inplace=True))

        def conv_dw(inp, oup, stride):
            return nn.Sequential(
                nn.Conv2d(inp, inp, 3, stride, 1, groups=inp, bias=False), nn.BatchNorm2d(inp), nn.ReLU(inplace=True),
                nn.Conv2d(inp, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup), nn.ReLU(inplace=True))

        c = self.base_channel
        self.model = nn.Sequential(
            conv_bn(3, c, 2),  # 160*120
            conv_dw(c, c * 2, 1),
            conv_dw(c * 2, c * 2, 2),  # 80*60
            conv_dw(c * 2, c * 2, 1),
            conv_dw(c * 2, c * 4, 2),  # 40*30
            conv_dw(c * 4, c * 4, 1),
            conv_dw(c * 4, c * 4, 1),
            BasicRFB(c * 4, c * 4, stride=1, scale=1.0),
            conv_dw(c * 4, c * 8, 2),  # 20*15
            conv_dw(c * 8, c * 8, 1),
            conv_dw(c * 8, c * 8, 1),
            conv_dw(c * 8, c * 16, 2),  # 10*8
            conv_dw(c * 16, c * 16, 1))


# ---- vision/ssd/mb_tiny_RFB_fd.py + vision/ssd/ssd.py
def SeperableConv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0):
    return nn.Sequential(
        nn.Conv2d(in_channels, in_channels, kernel_size=kernel_size, groups=in_channels, stride=stride, padding=padding),
        nn.ReLU(),
        nn.Conv2d(in_channels, out_channels, kernel_size=1))


MIN_BOXES = [[10, 16, 24], [32, 48], [64, 96], [128, 192, 256]]
STRIDES = [8, 16, 32, 64]
CENTER_VARIANCE, SIZE_VARIANCE = 0.1, 0.2


def generate_priors(image_size):
    """vision/utils/box_utils.py generate_priors + vision/ssd/config/fd_config.py define_img_size
    (float64 Python arithmetic, clamped to [0,1], float32 tensor)."""
    w, h = image_size
    feature_map_w_h_list = [[int(math.ceil(w / s)) for s in STRIDES], [int(math.ceil(h / s)) for s in STRIDES]]
    shrinkage_list = [[image_size[i] / fm for fm in feature_map_w_h_list[i]] for i in range(2)]
    priors = []
    for index in range(len(feature_map_w_h_list[0])):
        scale_w = image_size[0] / shrinkage_list[0][index]
        scale_h = image_size[1] / shrinkage_list[1][index]
        for j in range(feature_map_w_h_list[1][index]):
            for i in range(feature_map_w_h_list[0][index]):
                x_center = (i + 0.5) / scale_w
                y_center = (j + 0.5) / scale_h
                for min_box in MIN_BOXES[index]:
                    priors.append([x_center, y_center, min_box / image_size[0], min_box / image_size[1]])
    return torch.clamp(torch.tensor(priors), 0.0, 1.0)


class SSD(nn.Module):
    def __init__(self, image_size, num_classes=2):
        super().__init__()
        self.num_classes = num_classes
        base = Mb_Tiny_RFB(num_classes)
        self.base_net = base.model
        c = base.base_channel
        self.source_layer_indexes = [8, 11, 13]
        self.extras = nn.ModuleList([nn.Sequential(
            nn.Conv2d(in_channels=c * 16, out_channels=c * 4, kernel_size=1), nn.ReLU(),
            SeperableConv2d(in_channels=c * 4, out_channels=c * 16, kernel_size=3, stride=2, padding=1), nn.ReLU())])
        self.regression_headers = nn.ModuleList([
            SeperableConv2d(in_channels=c * 4, out_channels=3 * 4, kernel_size=3, padding=1),
            SeperableConv2d(in_channels=c * 8, out_channels=2 * 4, kernel_size=3, padding=1),
            SeperableConv2d(in_channels=c * 16, out_channels=2 * 4, kernel_size=3, padding=1),
            nn.Conv2d(in_channels=c * 16, out_channels=3 * 4, kernel_size=3, padding=1)])
        self.classification_headers = nn.ModuleList([
            SeperableConv2d(in_channels=c * 4, out_channels=3 * num_classes, kernel_size=3, padding=1),
            SeperableConv2d(in_channels=c * 8, out_channels=2 * num_classes, kernel_size=3, padding=1),
            SeperableConv2d(in_channels=c * 16, out_channels=2 * num_classes, kernel_size=3, padding=1),
            nn.Conv2d(in_channels=c * 16, out_channels=3 * num_classes, kernel_size=3, padding=1)])
        self.register_buffer("priors", generate_priors(image_size), persistent=False)

    def compute_header(self, i, x):
        confidence = self.classification_headers[i](x)
        confidence = confidence.permute(0, 2, 3, 1).contiguous()
        confidence = confidence.view(confidence.size(0), -1, self.num_classes)
        location = self.regression_headers[i](x)
        location = location.permute(0, 2, 3, 1).contiguous()
        location = location.view(location.size(0), -1, 4)
        return confidence, location

    def forward(self, x):
        confidences, locations = [], []
        start_layer_index, header_index = 0, 0
        for end_layer_index in self.source_layer_indexes:
            for layer in self.base_net[start_layer_index:end_layer_index]:
                x = layer(x)
            start_layer_index = end_layer_index
            confidence, location = self.compute_header(header_index, x)
            header_index += 1
            confidences.append(confidence)
            locations.append(location)
        for layer in self.base_net[end_layer_index:]:
            x = layer(x)
        for layer in self.extras:
            x = layer(x)
            confidence, location = self.compute_header(header_index, x)
            header_index += 1
            confidences.append(confidence)
            locations.append(location)
        confidences = torch.cat(confidences, 1)
        locations = torch.cat(locations, 1)
        # is_test branch of SSD.forward
        confidences = F.softmax(confidences, dim=2)
        priors = self.priors
        if priors.dim() + 1 == locations.dim():
            priors = priors.unsqueeze(0)
        boxes = torch.cat([locations[..., :2] * CENTER_VARIANCE * priors[..., 2:] + priors[..., :2],
                           torch.exp(locations[..., 2:] * SIZE_VARIANCE) * priors[..., 2:]], dim=locations.dim() - 1)
        boxes = torch.cat([boxes[..., :2] - boxes[..., 2:] / 2, boxes[..., :2] + boxes[..., 2:] / 2], boxes.dim() - 1)
        return confidences, boxes


def conv_modules(model):
    """The 52 Conv2d modules in execution order, each with its BatchNorm2d (or None)."""
    convs = []
    mods = []

    def walk(m):
        children = list(m.children())
        if not children:
            mods.append(m)
        for c in children:
            walk(c)

    # execution order of SSD.forward (heads interleaved with the backbone)
    order = []
    order += list(model.base_net[:8])
    order += [model.classification_headers[0], model.regression_headers[0]]
    order += list(model.base_net[8:11])
    order += [model.classification_headers[1], model.regression_headers[1]]
    order += list(model.base_net[11:13])
    order += [model.classification_headers[2], model.regression_headers[2]]
    order += [model.extras[0]]
    order += [model.classification_headers[3], model.regression_headers[3]]
    for blk in order:
        mods.clear()
        walk(blk)
        leaves = list(mods)
        for i, m in enumerate(leaves):
            if isinstance(m, nn.Conv2d):
                bn = leaves[i + 1] if i + 1 < len(leaves) and isinstance(leaves[i + 1], nn.BatchNorm2d) else None
                convs.append((m, bn))
    assert len(convs) == 52, len(convs)
    return convs


def build_seeded(image_size, seed=2025, cls_bias_shift=(-8.3, -4.1, -4.5, -5.7)):
    """The module with seeded He-normal conv weights and non-trivial BatchNorm statistics (so that
    BN folding is exercised), eval mode.  Face-class biases of the classification heads are shifted
    as in synth.CLS_BIAS_SHIFT so that a realistic fraction of priors passes the threshold."""
    g = torch.Generator().manual_seed(seed)
    model = SSD(image_size).eval()
    with torch.no_grad():
        for m, bn in conv_modules(model):
            fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
            if m.bias is not None:
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.01)
            if bn is not None:
                bn.weight.copy_(torch.rand(bn.weight.shape, generator=g) * 0.5 + 0.75)
                bn.bias.copy_(torch.randn(bn.bias.shape, generator=g) * 0.05)
                bn.running_mean.copy_(torch.randn(bn.running_mean.shape, generator=g) * 0.05)
                bn.running_var.copy_(torch.rand(bn.running_var.shape, generator=g) * 0.5 + 0.75)
        for h, shift in zip(model.classification_headers, cls_bias_shift):
            last = h if isinstance(h, nn.Conv2d) else h[2]
            last.bias[1::2] += shift
    return model


def folded_blob(model):
    """The packed blob the loader must produce: per conv w[cout][cin/g][k][k] then b[cout], BatchNorm
    folded in float64 (w' = w*g/sqrt(var+eps), b' = beta + (b-mean)*g/sqrt(var+eps))."""
    out = []
    for m, bn in conv_modules(model):
        w = m.weight.detach().double().numpy()
        b = m.bias.detach().double().numpy() if m.bias is not None else np.zeros(w.shape[0])
        if bn is not None:
            sc = bn.weight.detach().double().numpy() / np.sqrt(bn.running_var.detach().double().numpy() + bn.eps)
            w = w * sc[:, None, None, None]
            b = bn.bias.detach().double().numpy() + (b - bn.running_mean.detach().double().numpy()) * sc
        out += [w.ravel(), b.ravel()]
    return np.concatenate(out).astype(np.float32)


def export_onnx(model, image_size, path=None, fold_bn=False, opset=11):
    """torch.onnx.export (legacy TorchScript exporter: needs no `onnx` package for the export itself;
    its post-pass that only attaches onnxscript functions imports `onnx`, which is absent here and
    not needed, so that one hook is bypassed).  fold_bn=False keeps Conv + BatchNormalization pairs
    (TrainingMode.PRESERVE on an eval() module with the conv-bn peephole disabled by exporting BN
    in inference form); fold_bn=True lets the exporter fuse them, as a zoo re-export would."""
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils

    orig = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda proto, opsets: proto
    try:
        f = io.BytesIO()
        w, h = image_size
        x = torch.zeros(1, 3, h, w)
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(model, x, f, dynamo=False, opset_version=opset, input_names=["input"],
                              output_names=["scores", "boxes"], do_constant_folding=bool(fold_bn),
                              training=torch.onnx.TrainingMode.EVAL if fold_bn else torch.onnx.TrainingMode.PRESERVE)
    finally:
        onnx_proto_utils._add_onnxscript_fn = orig
    data = f.getvalue()
    if path:
        with open(path, "wb") as fp:
            fp.write(data)
    return data


def onnx_op_histogram(data):
    """Op types of the exported graph's nodes (hand-rolled protobuf walk; for test assertions)."""
    def varint(b, i):
        v = s = 0
        while True:
            c = b[i]
            i += 1
            v |= (c & 0x7F) << s
            s += 7
            if not c & 0x80:
                return v, i

    def fields(b):
        i = 0
        while i < len(b):
            key, i = varint(b, i)
            num, wire = key >> 3, key & 7
            if wire == 0:
                v, i = varint(b, i)
            elif wire == 2:
                n, i = varint(b, i)
                v = b[i:i + n]
                i += n
            elif wire == 5:
                v = b[i:i + 4]
                i += 4
            elif wire == 1:
                v = b[i:i + 8]
                i += 8
            else:
                raise ValueError(wire)
            yield num, wire, v

    hist = {}
    for num, wire, v in fields(data):
        if num == 7 and wire == 2:
            for n2, w2, node in fields(v):
                if n2 == 1 and w2 == 2:
                    for n3, w3, val in fields(node):
                        if n3 == 4:
                            op = bytes(val).decode()
                            hist[op] = hist.get(op, 0) + 1
    return hist


if __name__ == "__main__":
    import sys

    size = (640, 480) if len(sys.argv) < 2 or sys.argv[1] == "640" else (320, 240)
    m = build_seeded(size)
    for fold in (False, True):
        d = export_onnx(m, size, fold_bn=fold)
        print("fold_bn=%s: %d bytes, ops %s" % (fold, len(d), onnx_op_histogram(d)))
