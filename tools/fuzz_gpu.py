"""Robustness fuzz on the GPU box: random corruptions over whole JPEG files (headers included)
through ufd_infer_jpeg_batch and, every other round, ufd_annotate_jpeg_batch (rectangles + labels + re-encode of
whatever the damaged frame decodes to) in mixed batches.  Every frame must end as OK / UFD_E_TRUNCATED / UFD_E_DECODE /
UFD_E_UNSUPPORTED / UFD_E_TOO_LARGE, every annotated stream must be a framed JPEG, and a clean batch must still decode
bit-exactly afterwards.
Usage: python tools/fuzz_gpu.py [rounds] [big|model|layouts]   (exits non-zero on any violation; "big": 640x480-class frames through
UltraFace-640 instead of thumbnail-sized ones through UltraFace-320; "model": every frame exactly 640x480, so that the 4:2:0 and
4:2:2 ones take the fused stem -- sample planes straight into the first convolution -- in mixed, damaged batches; "layouts"
(round 6): the libjpeg-turbo-written streams of tests/golden/jpeg_layouts.npz as seeds -- 4:1:1, 4:1:0 with its ten blocks per
MCU, 4:4:0, RGB colour space, luma coarser than chroma, non-interleaved scans -- so that damaged sampling factors, scan scripts
and ten-block MCUs meet the device entropy decoder and the generic upsampler)"""
import sys
import numpy as np

sys.path.insert(0, ".")
from infercam_onnx_amd import nn, synth


def corrupt(rng, j):
    b = bytearray(j)
    kind = rng.integers(0, 6)
    if kind == 0:
        for p in rng.integers(2, min(len(b), 700), size=int(rng.integers(1, 4))):
            b[int(p)] ^= int(rng.integers(1, 256))
    elif kind == 1:
        for p in rng.integers(2, len(b), size=int(rng.integers(1, 6))):
            b[int(p)] ^= int(rng.integers(1, 256))
    elif kind == 2:
        b = b[:int(rng.integers(2, len(b)))]
    elif kind == 3:
        p = int(rng.integers(2, len(b) - 2))
        b[p:p + 2] = bytes([0xFF, int(rng.integers(0xC0, 0xFF))])
    elif kind == 4:
        p = int(rng.integers(2, len(b) - 40))
        b[p:p] = b[p:p + int(rng.integers(1, 40))]
    else:
        p = int(rng.integers(2, min(len(b), 700)))
        b[p:p + 2] = b"\x00\x00"
    return bytes(b)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(11)
    w = synth.synthetic_weights()
    model_size = len(sys.argv) > 2 and sys.argv[2] == "model"
    big = model_size or (len(sys.argv) > 2 and sys.argv[2] == "big")
    layouts = len(sys.argv) > 2 and sys.argv[2] == "layouts"
    if big:
        m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=w, priors=synth.gen_priors(640, 480), max_batch=8,
                              max_src=(704, 544), det_cap=17640)
    else:
        m = nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=w, priors=synth.gen_priors(320, 240), max_batch=8,
                              max_src=(640, 480), det_cap=4420)
    w0, h0 = (640, 480) if big else (96, 64)
    step = 0 if model_size else 8
    base = [synth.encode_jpeg(synth.synth_frame(5, i, w0 + step * i, h0 + step * i), **kw) for i, kw in enumerate((
        {}, {"restart_rows": 1}, {"subsampling": "4:2:2"}, {"progressive": True}, {"optimize": True, "quality": 30},
        {"subsampling": "4:4:4", "restart_rows": 2}, {"quality": 100}, {"subsampling": "4:2:2", "restart_rows": 1}))]
    if layouts:
        import os

        z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "jpeg_layouts.npz"))
        base = [z[k].tobytes() for k in ("411_150x100/base", "410_640x480/dri_row", "440_150x100/dri_3", "rgb_adobe0_150x100/base",
                                         "y11c22_150x100/base", "cb11cr22_150x100/nonint", "gray_sof22_67x45/base", "410_37x29/nonint_y_cc_dri")]
    ref, st = m.infer_jpeg_batch(base)
    assert st == [0] * len(base), st
    allowed = (0, nn.UFD_E_TRUNCATED, nn.UFD_E_DECODE, nn.UFD_E_UNSUPPORTED, nn.UFD_E_TOO_LARGE)
    seen = {}
    for r in range(rounds):
        batch = [corrupt(rng, j) if rng.random() < 0.8 else j for j in base]
        if r & 1:
            res, st, streams = m.annotate_jpeg_batch(batch, (1280, 720), quality=int(rng.integers(1, 101)))
            for s, x in zip(st, streams):
                if s in (0, nn.UFD_E_TRUNCATED) and not (x and x[:2] == b"\xff\xd8" and x[-2:] == b"\xff\xd9"):
                    print("annotated stream of an OK frame is not a JPEG")
                    sys.exit(4)
        else:
            res, st = m.infer_jpeg_batch(batch)
        for s in st:
            seen[s] = seen.get(s, 0) + 1
        if not all(s in allowed for s in st):
            print("unexpected status", st)
            sys.exit(2)
    again, st = m.infer_jpeg_batch(base)
    if st != [0] * len(base) or again != ref:
        print("clean batch differs after the fuzz rounds", st)
        sys.exit(3)
    print("fuzz ok:", rounds, "rounds, statuses", seen)


main()
