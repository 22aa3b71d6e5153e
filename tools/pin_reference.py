#!/usr/bin/env python3
"""One step from "parity unpinned" to pinned, for the day the zoo files can be had.

The reference's only pinned results are the face counts of its eight test pictures with UltraFace-640, conf > 0.5,
IoU <= 0.5 (infer_server/tests/integration_tests.rs:20-35: 3, 6, 4, 3, 1, 1, 10, 0), and they need the weights the
reference downloads at run time (nn.rs:21-22,155-162):
    version-RFB-640.onnx / version-RFB-320.onnx  of  onnx/models  vision/body_analysis/ultraface/models
Given those two files this script
  1. parses each with the product's loader (ufd_debug_load_onnx: host only) -- BatchNorm pairs folded, priors read from the file;
  2. runs the eight pictures (tests/golden/test_pics = the reference's resources/test_pics) through the CPU oracle at both
     variants and ASSERTS the eight counts for the 640 model -- this is what pins the oracle to the reference;
  3. with a GPU present, runs them through the product (ufd_infer_jpeg) and asserts it equals the oracle (1e-4, the tests' rule);
  4. writes tests/golden/reference_boxes.npz: per picture and variant the oracle's [n, 5] detections, plus the sha256 of
     the two files -- from then on tests/test_oracle_cnn.py::test_reference_boxes_if_pinned (CPU) and the two armed GPU tests
     (tests/test_gpu_mirrors.py) hold every later build to the reference's own numbers.

    python tools/pin_reference.py [--onnx-640 PATH] [--onnx-320 PATH] [--no-gpu]
Default paths: $XDG_CACHE_HOME or ~/.cache, /infercam_onnx/ultraface-RFB-{640,320}.onnx (the reference's cache, nn.rs:144-156).
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REFERENCE_COUNTS_MSG = "integration_tests.rs:20-29"


def cache_path(variant):
    return os.path.join(os.environ.get("XDG_CACHE_HOME", os.path.expanduser("~/.cache")), "infercam_onnx", "ultraface-RFB-%d.onnx" % variant)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--onnx-640", default=cache_path(640))
    ap.add_argument("--onnx-320", default=cache_path(320))
    ap.add_argument("--no-gpu", action="store_true", help="oracle only (steps 1, 2, 4)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "reference_boxes.npz"))
    args = ap.parse_args()

    import oracle
    from infercam_onnx_amd import nn, synth

    oracle.build()
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "test_pics.json")))
    pics = {f: open(os.path.join(ROOT, "tests", "golden", "test_pics", f), "rb").read() for f in meta}
    out, failed = {}, []
    for variant, path in ((640, args.onnx_640), (320, args.onnx_320)):
        if not os.path.exists(path):
            print("UltraFace-%d: %s absent -- skipped" % (variant, path))
            continue
        W, H = (640, 480) if variant == 640 else (320, 240)
        weights, priors = nn.load_onnx(path, variant)  # raises with the loader's message on a graph it does not know
        if priors is None:
            print("UltraFace-%d: no priors constant in the file, regenerated (SURVEY 8.1)" % variant)
            priors = synth.gen_priors(W, H)
        out["sha256_%d" % variant] = np.frombuffer(hashlib.sha256(open(path, "rb").read()).digest(), np.uint8)
        refs = {}
        for f, jpeg in pics.items():
            refs[f] = np.asarray(oracle.infer_jpeg(jpeg, W, H, weights, priors, 0.5, 0.5), np.float32).reshape(-1, 5)
            out["%d/%s" % (variant, f)] = refs[f]
            line = "UltraFace-%d %-28s oracle %2d faces" % (variant, f, len(refs[f]))
            if variant == 640:
                want = meta[f]["reference_face_count"]
                line += "   reference %2d (%s)%s" % (want, REFERENCE_COUNTS_MSG, "" if want == len(refs[f]) else "   <-- MISMATCH")
                if want != len(refs[f]):
                    failed.append("oracle: %s %d != %d" % (f, len(refs[f]), want))
            print(line)
        import torch

        if not args.no_gpu and torch.cuda.device_count() > 0:
            from helpers import assert_dets_match, dets_array

            v = nn.UltrafaceVariant.W640H480 if variant == 640 else nn.UltrafaceVariant.W320H240
            with nn.UltrafaceModel(v, 0.5, 0.5, weights_path=path, max_src=(1280, 1024)) as m:
                for f, jpeg in pics.items():
                    got = dets_array(m.infer_jpeg(jpeg))
                    try:
                        assert_dets_match(got, refs[f], what="GPU vs oracle, %s" % f)
                    except AssertionError as e:
                        failed.append(str(e).splitlines()[0])
                    if variant == 640 and len(got) != meta[f]["reference_face_count"]:
                        failed.append("GPU: %s %d != %d" % (f, len(got), meta[f]["reference_face_count"]))
            print("UltraFace-%d: GPU path checked against the oracle on %d pictures" % (variant, len(pics)))
        else:
            print("UltraFace-%d: GPU leg not run" % variant)
    if failed:
        print("NOT PINNED:\n  " + "\n  ".join(failed))
        return 1
    if not any(k.startswith("sha256_") for k in out):
        print("nothing to pin: neither file is present")
        return 2
    np.savez_compressed(args.out, **out)
    if "sha256_640" not in out:
        print("wrote %s, but NOTHING IS PINNED by it: the reference's eight counts are for the 640 model" % args.out)
        return 3
    print("wrote %s -- the oracle reproduces the reference's own vector (%s)" % (args.out, REFERENCE_COUNTS_MSG))
    return 0


if __name__ == "__main__":
    sys.exit(main())
