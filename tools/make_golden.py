"""Generates the committed golden fixtures under tests/golden/ (run from the repo root, in the
build container).  Sources of truth, none of them this repo's own code:
  * JPEG decode: libjpeg-turbo via PIL (the library behind the reference's turbojpeg crate,
    same defaults as tjDecompress2(flags=0)).
  * test_pics/*.jpg: the reference's own test pictures (data files held by
    infer_server/tests/integration_tests.rs:20-29); expected RGB = PIL decode, stored as sha256.
  * CNN: torch.nn.functional.conv2d (float64) on the seeded synthetic weights.
"""
import hashlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image  # noqa: E402

from infercam_onnx_amd import synth, topology as T  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def pil_rgb(b):
    return np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))


def jpeg_fixtures():
    cases = {}
    specs = [("base420_64x48", (64, 48), dict(subsampling="4:2:0")),
             ("base422_rst_65x47", (65, 47), dict(subsampling="4:2:2", restart_rows=1)),
             ("base444_33x17", (33, 17), dict(subsampling="4:4:4")),
             ("prog420_80x56", (80, 56), dict(subsampling="4:2:0", progressive=True)),
             ("base420_q30_96x64", (96, 64), dict(subsampling="4:2:0", quality=30))]
    for name, (w, h), kw in specs:
        jpeg = synth.encode_jpeg(synth.synth_frame(77, len(cases), w, h), **kw)
        cases[name + "_jpeg"] = np.frombuffer(jpeg, np.uint8)
        cases[name + "_rgb"] = pil_rgb(jpeg)
    np.savez_compressed(os.path.join(G, "jpeg_small.npz"), **cases)
    pics = {}
    for f in sorted(os.listdir(os.path.join(G, "test_pics"))):
        b = open(os.path.join(G, "test_pics", f), "rb").read()
        rgb = pil_rgb(b)
        pics[f] = {"shape": list(rgb.shape), "sha256": hashlib.sha256(rgb.tobytes()).hexdigest()}
    # face counts pinned by the reference's only model test (integration_tests.rs:20-29)
    counts = {"bruce-mars-ZXq7xoo98b0-unsplash.jpg": 3, "clarke-sanders-ybPJ47PMT_M-unsplash.jpg": 6,
              "helena-lopes-e3OUQGT9bWU-unsplash.jpg": 4, "kaleidico-d6rTXEtOclk-unsplash.jpg": 3,
              "michael-dam-mEZ3PoFGs_k-unsplash.jpg": 1, "mika-W0i1N6FdCWA-unsplash.jpg": 1,
              "omar-lopez-T6zu4jFhVwg-unsplash.jpg": 10, "ken-cheung-KonWFWUaAuk-unsplash.jpg": 0}
    for f in pics:
        pics[f]["reference_face_count"] = counts[f]
    json.dump(pics, open(os.path.join(G, "test_pics.json"), "w"), indent=1, sort_keys=True)


def cnn_fixture():
    import torch
    import torch.nn.functional as F

    W, H = 320, 240
    blob = synth.synthetic_weights()
    pri = synth.gen_priors(W, H)
    frame = synth.synth_frame(synth.DEFAULT_FRAME_SEED, 0, W, H)
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    x = ((frame.astype(np.float32) / np.float32(255.0) - mean) / std).transpose(2, 0, 1).copy()
    acts = {}
    xin = torch.from_numpy(x)[None].double()
    for i, s in enumerate(T.CONVS):
        inp = xin if s.src == -1 else (torch.cat([acts[15], acts[18], acts[22]], 1) if s.src == -2 else acts[s.src])
        w, b = synth.layer_params(blob, i)
        y = F.conv2d(inp, torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=s.stride, padding=s.pad,
                     dilation=s.dil, groups=s.groups)
        if i == 24:
            y = torch.relu(acts[23] + y)
        elif s.relu:
            y = torch.relu(y)
        acts[i] = y
    cls = torch.cat([acts[l].permute(0, 2, 3, 1).reshape(1, -1, 2) for l in T.CLS_LAYERS], 1)
    reg = torch.cat([acts[l].permute(0, 2, 3, 1).reshape(1, -1, 4) for l in T.REG_LAYERS], 1)
    scores = torch.softmax(cls, 2)[0]
    p = torch.from_numpy(pri).double()
    c = reg[0, :, :2] * 0.1 * p[:, 2:] + p[:, :2]
    sz = torch.exp(reg[0, :, 2:] * 0.2) * p[:, 2:]
    boxes = torch.cat([c - sz / 2, c + sz / 2], 1)
    np.savez_compressed(os.path.join(G, "cnn_320.npz"), frame=frame, scores=scores.numpy().astype(np.float32),
                        boxes=boxes.numpy().astype(np.float32),
                        layer_absmax=np.array([float(acts[i].abs().max()) for i in range(T.NUM_CONV)]),
                        layer_mean=np.array([float(acts[i].mean()) for i in range(T.NUM_CONV)]))


if __name__ == "__main__":
    jpeg_fixtures()
    cnn_fixture()
    print("golden fixtures written to", G)
