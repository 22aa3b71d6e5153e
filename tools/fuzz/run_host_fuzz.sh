#!/bin/bash
# ASan + UBSan fuzz of the host-side parsers (CPU only; GPU sanitizers are not available on the pool).
#   tools/fuzz/run_host_fuzz.sh [rounds per seed, default 200] [rng seed]
# Seeds: synthetic camera-like JPEGs in every flavour the decoder handles (baseline / DRI / progressive / optimised /
# 4:2:2 / 4:4:4 / grey), the reference's own test pictures, and an UltraFace-RFB .onnx written by synth.write_onnx.
set -eu
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
work=${TMPDIR:-/tmp}/ufd_fuzz_$$
mkdir -p "$work/corpus"
trap 'rm -rf "$work"' EXIT
python3 - "$root" "$work/corpus" <<'PY'
import glob, os, shutil, sys
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from infercam_onnx_amd import synth
n = 0
for kw in ({}, {"restart_rows": 1}, {"progressive": True}, {"optimize": True}, {"subsampling": "4:2:2"}, {"subsampling": "4:4:4"}, {"quality": 30}):
    for (w, h) in ((64, 48), (160, 120), (33, 17)):
        open(os.path.join(out, "s%02d.jpg" % n), "wb").write(synth.encode_jpeg(synth.synth_frame(5, n, w, h), **kw))
        n += 1
from PIL import Image
import io
b = io.BytesIO(); Image.fromarray(synth.synth_frame(5, 99, 72, 40)).convert("L").save(b, "JPEG"); open(os.path.join(out, "grey.jpg"), "wb").write(b.getvalue())
for p in sorted(glob.glob(os.path.join(root, "tests", "golden", "test_pics", "*.jpg")))[:3]:
    shutil.copy(p, out)
synth.write_onnx(os.path.join(out, "model.onnx"), synth.synthetic_weights(), 320, 240, with_batchnorm=True)
PY
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer \
    -I"$root/infercam_onnx_amd/csrc" "$here/fuzz_host.cpp" "$root/infercam_onnx_amd/csrc/jpeg_host.cpp" \
    "$root/infercam_onnx_amd/csrc/onnx_loader.cpp" -o "$work/fuzz_host"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 "$work/fuzz_host" "$work/corpus" "${1:-200}" "${2:-1}"
