"""Host-side logic of the product library that needs no GPU: the C-ABI surface, the host Huffman
stage, the ONNX loader, the Python mirror of nn.rs, and loud failure without a device."""
import os
import subprocess
import sys
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from infercam_onnx_amd import nn

    lib = nn.load_library()
    header = open(os.path.join(ROOT, "include", "ufd.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void\*?|size_t|uint32_t|const char\*)\s+(ufd_\w+)\s*\(", header, re.M))
    assert declared == set(nn.ABI_SYMBOLS), declared ^ set(nn.ABI_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_no_oracle_in_product():
    """The product must not include, link or import the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "infercam_onnx_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", "Makefile")):
                src = open(os.path.join(dirpath, f)).read()
                assert "ufd_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


@pytest.mark.parametrize("kw", [{}, {"restart_rows": 1}, {"progressive": True}, {"optimize": True},
                                {"subsampling": "4:2:2"}, {"subsampling": "4:4:4"}, {"quality": 30}])
@pytest.mark.parametrize("size", [(640, 480), (65, 47), (1, 1)])
def test_host_huffman_coefficients_match_oracle(oracle_lib, kw, size):
    import ctypes
    from infercam_onnx_amd import nn, synth

    jpeg = synth.encode_jpeg(synth.synth_frame(13, size[0], size[0], size[1]), **kw)
    coef, w, h = nn.jpeg_coefficients(jpeg)
    L = oracle_lib.lib()
    L.ufo_jpeg_coefficients.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_size_t)]
    buf = (ctypes.c_char * len(jpeg)).from_buffer_copy(jpeg)
    n = ctypes.c_size_t()
    assert L.ufo_jpeg_coefficients(buf, len(jpeg), None, 0, ctypes.byref(n)) == 0
    ref = np.empty(n.value, np.int16)
    assert L.ufo_jpeg_coefficients(buf, len(jpeg), ref.ctypes.data, ref.size, ctypes.byref(n)) == 0
    assert (w, h) == size and np.array_equal(coef, ref)


def test_host_huffman_reference_test_pics(oracle_lib):
    import ctypes
    from infercam_onnx_amd import nn

    d = os.path.join(ROOT, "tests", "golden", "test_pics")
    L = oracle_lib.lib()
    L.ufo_jpeg_coefficients.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_size_t)]
    for f in sorted(os.listdir(d))[:3]:
        jpeg = open(os.path.join(d, f), "rb").read()
        coef, w, h = nn.jpeg_coefficients(jpeg)
        buf = (ctypes.c_char * len(jpeg)).from_buffer_copy(jpeg)
        n = ctypes.c_size_t(coef.size)
        ref = np.empty(coef.size, np.int16)
        assert L.ufo_jpeg_coefficients(buf, len(jpeg), ref.ctypes.data, ref.size, ctypes.byref(n)) == 0
        assert np.array_equal(coef, ref), f


def test_corrupt_jpeg_status_codes():
    from infercam_onnx_amd import nn, synth

    jpeg = synth.encode_jpeg(synth.synth_frame(1, 1, 64, 48))
    for bad in (jpeg[:300], jpeg[:-2], b"\xff\xd8\xff", b"junk"):
        with pytest.raises(nn.UfdError) as e:
            nn.jpeg_coefficients(bad)
        assert e.value.code == nn.UFD_E_DECODE
    # arithmetic-coded / lossless SOF markers are "unsupported", not "corrupt"
    sof9 = jpeg.replace(b"\xff\xc0", b"\xff\xc9", 1)
    with pytest.raises(nn.UfdError) as e:
        nn.jpeg_coefficients(sof9)
    assert e.value.code == nn.UFD_E_UNSUPPORTED
    # four components (CMYK / YCCK): libjpeg-turbo decodes them, but not to RGB -- tjDecompress2 into TJPF_RGB fails and the
    # reference's `expect` panics (inferer.rs:35-36); here: a status, and the oracle refuses too
    import io

    import numpy as np
    from PIL import Image

    import oracle

    buf = io.BytesIO()
    Image.fromarray(np.random.default_rng(0).integers(0, 255, (24, 40, 4), dtype=np.uint8), "CMYK").save(buf, "JPEG")
    with pytest.raises(nn.UfdError) as e:
        nn.jpeg_coefficients(buf.getvalue())
    assert e.value.code == nn.UFD_E_UNSUPPORTED
    with pytest.raises(oracle.OracleError):
        oracle.jpeg_decode_rgb(buf.getvalue())


def test_onnx_loader_roundtrip_and_bn_folding(tmp_path, weights):
    from infercam_onnx_amd import nn, synth

    for variant, (w, h) in ((640, (640, 480)), (320, (320, 240))):
        for bn in (False, True):
            for pri_as in ("constant", "initializer", "none"):
                p = str(tmp_path / ("m_%d_%d_%s.onnx" % (variant, bn, pri_as)))
                expect = synth.write_onnx(p, weights, w, h, with_batchnorm=bn, priors_as=pri_as)
                blob, pri = nn.load_onnx(p, variant)
                assert np.allclose(blob, expect, rtol=1e-6, atol=1e-7)
                if bn:
                    assert not np.array_equal(expect, weights)
                if pri_as == "none":
                    assert pri is None
                else:
                    assert np.array_equal(pri, synth.gen_priors(w, h))


def test_onnx_loader_rejects_other_graphs(tmp_path, weights):
    from infercam_onnx_amd import nn, synth

    p = str(tmp_path / "m.onnx")
    synth.write_onnx(p, weights, 640, 480)
    data = open(p, "rb").read()
    with pytest.raises(nn.UfdError) as e:
        nn.load_onnx(str(tmp_path / "missing.onnx"), 640)
    assert e.value.code == nn.UFD_E_WEIGHTS
    open(p, "wb").write(data[: len(data) // 2])
    with pytest.raises(nn.UfdError):
        nn.load_onnx(p, 640)
    open(p, "wb").write(b"\x08\x04")  # a ModelProto without a graph
    with pytest.raises(nn.UfdError):
        nn.load_onnx(p, 640)


def test_variant_mirror_and_loud_failure_without_gpu(weights):
    import torch
    from infercam_onnx_amd import nn

    assert nn.UltrafaceVariant.W640H480.width_height() == (640, 480)  # nn.rs:36-41
    assert nn.UltrafaceVariant.W320H240.width_height() == (320, 240)
    if torch.cuda.device_count() == 0:
        with pytest.raises(nn.UfdError) as e:
            nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=weights)
        assert e.value.code == nn.UFD_E_DEVICE and "no CPU fallback" in str(e.value)


def test_experiment_knobs_in_the_environment_are_refused_not_obeyed(weights):
    """The measurement knobs of docs/EXPERIMENTS.md (UFD_ABLATE_LAYERS skips layers, UFD_REPEAT_ENTROPY / UFD_EXTEND_ROUNDS
    change the decode chain ...) exist only in the `make EXPERIMENTS=1` build; the library that ships refuses to create a
    handle while one is set -- it used to obey them silently, so a variable left in a server's environment changed results
    with every call still returning UFD_OK (round-5 advisor finding)."""
    import ctypes
    from infercam_onnx_amd import nn

    L = nn.load_library()
    for knob in ("UFD_ABLATE_LAYERS", "UFD_REPEAT_ENTROPY", "UFD_EXTEND_ROUNDS", "UFD_SUB_SMALL_BYTES", "UFD_TEST_DUPLICATE_DEVICES"):
        os.environ[knob] = "1"
        try:
            with pytest.raises(nn.UfdError) as e:
                nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=weights)
            assert e.value.code == nn.UFD_E_ARG and knob in str(e.value) and "EXPERIMENTS=1" in str(e.value)
            cfg, keep = nn.UltrafaceModel._config(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=weights)
            out = (ctypes.c_void_p * 2)()
            assert L.ufd_create_replicas(ctypes.byref(cfg), (ctypes.c_int32 * 1)(0), 1, out) == nn.UFD_E_ARG
            assert knob in (L.ufd_last_error(None) or b"").decode() and out[0] is None
        finally:
            del os.environ[knob]
    src = "".join(open(os.path.join(ROOT, "infercam_onnx_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "infercam_onnx_amd", "csrc"))
                  if f.endswith((".cpp", ".hip", ".hpp")) and f != "experiments.hpp")
    import re
    assert set(re.findall(r'getenv\("(\w+)"\)', src)) <= {"XDG_CACHE_HOME", "HOME"}, "a knob outside experiments.hpp"


def test_prime_device_argument_and_error_paths():
    """ufd_prime_device (a Python host with torch calls it before torch touches the GPU: include/ufd.h): without a GPU it
    says so like ufd_create does; with one, a bad device id is UFD_E_ARG."""
    import torch
    from infercam_onnx_amd import nn

    L = nn.load_library()
    if torch.cuda.device_count() == 0:
        assert L.ufd_prime_device(0) == nn.UFD_E_DEVICE and b"no CPU fallback" in (L.ufd_last_error(None) or b"")
        with pytest.raises(nn.UfdError):
            nn.prime_device(0)
    else:
        assert L.ufd_prime_device(0) == 0 and L.ufd_prime_device(99) == nn.UFD_E_ARG and L.ufd_prime_device(-1) == nn.UFD_E_ARG


def test_synthetic_inputs_are_deterministic():
    from infercam_onnx_amd import synth

    a, b = synth.synthetic_weights(), synth.synthetic_weights()
    assert np.array_equal(a, b) and a.size == 273888
    assert np.array_equal(synth.synth_frame(1, 2, 64, 48), synth.synth_frame(1, 2, 64, 48))
    assert not np.array_equal(synth.synth_frame(1, 2, 64, 48), synth.synth_frame(1, 3, 64, 48))
    j = synth.encode_jpeg(synth.synth_frame(synth.DEFAULT_FRAME_SEED, 0, 640, 480))
    assert 15000 < len(j) < 90000  # camera-like size, not incompressible noise


def test_inferer_loop_pipelining_and_ordering():
    """Host logic of the Inferer mirror (inferer.rs:29-50) with a stand-in model: results reach the
    slots in order, at most `depth` batches are in flight, and a lone frame is delivered without
    waiting for the next one to arrive."""
    import queue
    import threading

    from infercam_onnx_amd.inferer import Inferer

    class FakeModel:
        def __init__(self):
            self.in_flight = 0
            self.max_in_flight = 0
            self.batches = []

        def submit_jpeg_batch(self, jpegs):
            self.in_flight += 1
            self.max_in_flight = max(self.max_in_flight, self.in_flight)
            self.batches.append(list(jpegs))
            return len(self.batches) - 1

        def wait(self, ticket):
            self.in_flight -= 1
            jpegs = self.batches[ticket]
            return [[("det", j)] for j in jpegs], [0] * len(jpegs)

    # a burst: 50 frames queued up front, batches of 4, depth 3
    rx, got, model = queue.Queue(), [], FakeModel()
    for i in range(50):
        rx.put((1280, 720, b"frame%d" % i, lambda r, i=i: got.append((i, r))))
    rx.put(None)
    Inferer(rx, model=model, max_batch=4, depth=3).run()
    assert [i for i, _ in got] == list(range(50))
    assert all(r == ([("det", b"frame%d" % i)], 0) for i, r in got)
    assert 2 <= model.max_in_flight <= 3 and model.in_flight == 0
    assert all(len(b) <= 4 for b in model.batches)

    # a lone frame: its result arrives while the loop is still waiting for the next slot
    rx, model, delivered = queue.Queue(), FakeModel(), threading.Event()
    rx.put((1280, 720, b"only", lambda r: delivered.set()))
    t = threading.Thread(target=Inferer(rx, model=model, max_batch=4).run, daemon=True)
    t.start()
    assert delivered.wait(5.0)
    rx.put(None)
    t.join(5.0)
    assert not t.is_alive()


def test_host_parsers_clean_under_asan_ubsan():
    """tools/fuzz/run_host_fuzz.sh: the JPEG header/marker scanner + entropy decoder and the ONNX reader, built with
    -fsanitize=address,undefined, on mutated seeds of every stream flavour (a short campaign here; run the script with
    more rounds for a long one).  Any memory error or undefined behaviour aborts it."""
    import shutil
    import subprocess

    if not shutil.which("g++"):
        pytest.skip("no g++")
    r = subprocess.run([os.path.join(ROOT, "tools", "fuzz", "run_host_fuzz.sh"), "30", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "clean" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_create_replicas_argument_and_error_paths(weights):
    """ufd_create_replicas (one process, one handle per GPU, RCCL broadcast of the packed weights): every refusal leaves
    out[] NULL and says why.  Argument checks come before any device or RCCL work, so they run without a GPU."""
    import ctypes
    import torch
    from infercam_onnx_amd import nn

    L = nn.load_library()
    V = nn.UltrafaceVariant.W320H240

    def call(ids, n=None, cfg_mut=None, pass_cfg=True, pass_out=True):
        cfg, keep = nn.UltrafaceModel._config(V, 0.5, 0.5, weights=weights)
        if cfg_mut:
            cfg_mut(cfg)
        arr = (ctypes.c_int32 * max(len(ids), 1))(*ids)
        out = (ctypes.c_void_p * 70)(*([0xDEAD] * 70))
        rc = L.ufd_create_replicas(ctypes.byref(cfg) if pass_cfg else None, arr, len(ids) if n is None else n,
                                   out if pass_out else None)
        return rc, (L.ufd_last_error(None) or b"").decode(), out

    rc, msg, _ = call([0], pass_cfg=False)
    assert rc == nn.UFD_E_ARG and "null" in msg
    rc, msg, _ = call([0], pass_out=False)
    assert rc == nn.UFD_E_ARG
    rc, msg, out = call([0], n=0)
    assert rc == nn.UFD_E_ARG and "1..64" in msg
    rc, msg, out = call([0] * 65)
    assert rc == nn.UFD_E_ARG and "1..64" in msg
    rc, msg, out = call([0, 1, 0])
    assert rc == nn.UFD_E_ARG and "listed twice" in msg and all(out[i] is None for i in range(3))
    rc, msg, _ = call([0], cfg_mut=lambda c: setattr(c, "variant", 512))
    assert rc == nn.UFD_E_ARG and "variant" in msg
    rc, msg, _ = call([0], cfg_mut=lambda c: setattr(c, "struct_size", 8))
    assert rc == nn.UFD_E_ARG and "struct_size" in msg
    if torch.cuda.device_count() == 0:
        rc, msg, out = call([0, 1])
        assert rc == nn.UFD_E_DEVICE and "no CPU fallback" in msg and out[0] is None and out[1] is None
        with pytest.raises(nn.UfdError) as e:
            nn.UltrafaceModel.create_replicas(V, 0.5, 0.5, [0, 1], weights=weights)
        assert e.value.code == nn.UFD_E_DEVICE
    assert L.ufd_annotate_parity(0) == nn.UFD_PARITY_LABELS_UNPINNED
    assert L.ufd_annotate_parity(nn.UFD_ANNOT_NO_TEXT | nn.UFD_ANNOT_MULTIPART) == nn.UFD_PARITY_EXACT


# ---- the launch planner alone (ufd_debug_plan: no GPU) ----
_PLAN_FLAGS = {"keep_layers": 1, "no_chain": 32, "no_rfb_sum": 64, "no_dual": 512, "no_rfb_tail": 2048}


@pytest.mark.parametrize("variant", [320, 640])
@pytest.mark.parametrize("flags", [0, 32, 64, 512, 2048, 32 | 64, 64 | 2048, 32 | 512 | 2048, 1])
def test_planner_never_lets_two_live_tensors_share_arena_bytes(variant, flags):
    """The arena recycles an activation buffer behind its last reader.  For every combination of the plan flags: tensors
    whose arena ranges overlap have disjoint lifetimes (first write .. last read, in layer turns), every stored tensor lies
    inside the arena, a tensor some launch reads has storage, and a tensor without storage is written by no launch.  (Round
    4 found the one way this can go wrong on the GPU: the RFB concat under k_rfb_tail has no storage, and "giving it back"
    behind its last reader handed live bytes to the next tensor -- detections of one frame per batch changed.)"""
    from infercam_onnx_amd import nn

    for batch in (1, 5, 32):
        layers, tensors, arena = nn.debug_plan(variant, batch, flags)
        assert len(layers) == 52
        stored = [(i, t) for i, t in enumerate(tensors) if t["stored"]]
        for i, t in stored:
            assert t["size_floats"] == t["c"] * t["h"] * t["w"] * batch
            assert t["off_floats"] + t["size_floats"] <= arena, (i, t, arena)
            assert t["first"] < 52 and t["last"] >= t["first"] or t["last"] == -1, (i, t)
        for a in range(len(stored)):
            for b in range(a + 1, len(stored)):
                (ia, ta), (ib, tb) = stored[a], stored[b]
                overlap = ta["off_floats"] < tb["off_floats"] + tb["size_floats"] and tb["off_floats"] < ta["off_floats"] + ta["size_floats"]
                if not overlap:
                    continue
                if flags & 1:  # keep_layers: nothing is recycled
                    raise AssertionError("keep_layers plan: tensors %d and %d overlap" % (ia, ib))
                # a buffer is reused only by a tensor first written AFTER the turn of its last reader
                la, lb = max(ta["last"], ta["first"]), max(tb["last"], tb["first"])
                assert la < tb["first"] or lb < ta["first"], "tensors %d %s and %d %s are live together on the same bytes" % (ia, ta, ib, tb)
        for li, L in enumerate(layers):
            out = tensors[L["out_tensor"]]
            computed_elsewhere = L["chained"] or (L["kind"] == 3 and not L["materialize"])
            if not computed_elsewhere:
                assert out["stored"], (li, L, out)  # what a launch writes has storage ...
                if L["in_tensor"] >= 0:
                    assert tensors[L["in_tensor"]]["stored"] or L["kind"] in (1, 2), (li, L)  # ... and so has what it reads
            assert L["launches"] == int(not computed_elsewhere and L["leader"] == li and L["ride"] < 0)
        launches = sum(L["launches"] for L in layers)
        # 640: stem; m1->m2; m3->m4; m5; m6; RFB reduce stack, first 3x3s, b2 middle, tail; heads 0 (m8 rides); m9; m10; heads 1
        # (m11 rides); m12; heads 2 (extra.0 rides); extra.2 dw; extra.2 pw; heads 3.  320: its 8x10 maps (W = 10 is not a
        # multiple of 4) keep m11 ... heads 2 as depthwise + pointwise launches.
        # (k_rfb_tail from four 60x80 maps' worth of pixels on: a handle for a frame or two keeps the two launches)
        tail = flags == 0 and batch * (4800 if variant == 640 else 1200) >= 4 * 4800
        if flags == 0:
            assert launches == (19 if variant == 640 else 26) - int(tail) and sum(L["rfb_tail"] for L in layers) == int(tail), launches
            gone = [li for li, L in enumerate(layers) if L["tap_tensor"] < 0]
            for li in (2, 6, 23) + ((15, 18, 22) if tail else ()):  # m1.pw, m3.pw (chained), rfb.linear (summed); the dilated RFB convs inside k_rfb_tail
                assert li in gone, (li, gone)
        if flags == 2048:
            assert launches == (19 if variant == 640 else 26) and not any(L["rfb_tail"] for L in layers)


def test_detection_list_comparison_rule():
    """oracle/compare.py (round 5): the one rule by which the tests AND bench.py's `verified` block compare a frame's detections
    with the oracle's -- equal within the tolerance in order; else matched as sets, and a detection without a partner is excused
    only when its decision provably sat on the confidence threshold (strict `>`, nn.rs:121-128) or on max_iou (strict `>`,
    nn.rs:209-214).  bench.py fails on any other leftover (it used to tolerate one mismatching frame in 64)."""
    from oracle.compare import match_detections

    a = np.array([[0.1, 0.1, 0.3, 0.3, 0.9], [0.5, 0.5, 0.7, 0.7, 0.8]], np.float32)
    r = match_detections(a, a + 5e-5)
    assert r["equal"] and not r["left_got"] and not r["not_borderline"]
    # two detections whose confidences differ by less than the tolerance swap places: equal as sets
    b = np.array([[0.5, 0.5, 0.7, 0.7, 0.80004], [0.1, 0.1, 0.3, 0.3, 0.80001]], np.float32)
    r = match_detections(b, b[::-1].copy())
    assert not r["equal"] and not r["left_got"] and not r["left_ref"] and r["max_err"] <= 1e-4
    # an extra detection whose confidence sits on the threshold: unmatched but borderline
    c = np.vstack([a, [[0.8, 0.8, 0.9, 0.9, 0.50003]]]).astype(np.float32)
    r = match_detections(c, a)
    assert len(r["left_got"]) == 1 and not r["not_borderline"]
    # an extra detection well above the threshold and overlapping nothing: a real disagreement
    d = np.vstack([a, [[0.8, 0.8, 0.9, 0.9, 0.7]]]).astype(np.float32)
    r = match_detections(d, a)
    assert len(r["left_got"]) == 1 and len(r["not_borderline"]) == 1
    # ... unless its IoU with a kept detection of the other list is within 1e-3 of max_iou (the NMS decision flipped)
    e = np.vstack([a, [[0.1, 0.1, 0.3, 0.5, 0.7]]]).astype(np.float32)  # IoU with a[0]: 0.04 / 0.08 = 0.5
    r = match_detections(e, a, max_iou=0.5)
    assert len(r["left_got"]) == 1 and not r["not_borderline"]
    assert match_detections(np.zeros((0, 5), np.float32), np.zeros((0, 5), np.float32))["equal"]


def _built_kernel_objects():
    """The library's object files (csrc/build/*_kernels.o: hipcc cross-compiles them without a GPU); built here if a fresh
    check-out has none yet."""
    import glob

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = os.path.join(root, "infercam_onnx_amd", "csrc", "build", "*_kernels.o")
    if not glob.glob(pat):
        subprocess.run(["make", "-C", os.path.join(root, "infercam_onnx_amd", "csrc"), "-j8"], check=True, capture_output=True, timeout=1500)
    assert glob.glob(pat)


def test_prefetch_queues_are_in_the_isa():
    """tools/ab/r5_queue_audit.py on the built objects: the software prefetch of the kernels round 5 repaired is still one in
    the code hipcc emits -- no `vmcnt(0)` (a wait for the load that was just requested) in the k-loops of the 1x1 kernel and
    of the chained dw->pw kernels, and the queue depths the source asks for (rules 6 and 7).  A compiler or
    source change that lets the loads sink again shows here before it shows as 10-25 % on those kernels."""
    import re

    _built_kernel_objects()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ab", "r5_queue_audit.py"), "k_pw_mfma<1, 4, 1>", "k_dwpw2_mfma<32, 2, false>",
                        "k_dwpw2_mfma<16, 1, true>", "k_conv3x3_rows_mfma<1, 1>"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    loops = {}
    lines = r.stdout.splitlines()
    for head, body in zip(lines[0::2], lines[1::2]):
        loops[head.split(" (hot loop")[0]] = body.strip()
    waits = lambda k: [int(x) for x in re.findall(r"\((\d+)\)", loops[k])]
    assert set(waits("k_pw_mfma<1, 4, 1>")) == {3}, loops["k_pw_mfma<1, 4, 1>"]              # four k-steps of activations in flight
    w = waits("k_dwpw2_mfma<32, 2, false>")
    assert min(w[2:]) >= 6 and len(w) > 40, loops["k_dwpw2_mfma<32, 2, false>"]             # behind the row's first step: two windows (6 rows) stay in flight
    w = waits("k_dwpw2_mfma<16, 1, true>")
    assert w[1:] == [4, 4, 4, 4], loops["k_dwpw2_mfma<16, 1, true>"]                           # k-steps 4..7 of an X1 row: four requests behind the one consumed
    assert waits("k_conv3x3_rows_mfma<1, 1>")[:3] == [5, 4, 3], loops["k_conv3x3_rows_mfma<1, 1>"]  # the next chunk's rows stay in flight


def test_isa_mix_reads_the_built_code_objects(tmp_path):
    """tools/isa_mix.py (round 5's instruction ledger) on the library's own object files: every MFMA kernel instance is found
    with its hot loop, the chained kernel's k-loop carries its DPP multiply-adds and no select, and the JSON beside the table
    is what tools/design_table.py reads."""
    import json

    _built_kernel_objects()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "isa_mix.txt")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_mix.py"), "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    inst = json.load(open(str(tmp_path / "isa_mix.json")))["instances"]
    for k in ("k_dwpw2_mfma<16, 1, true>", "k_dwpw2_mfma<32, 2, false>", "k_dwpw_mfma<1, 1, 2, 1>", "k_dwpw_coop<1, 4>", "k_rfb_tail",
              "k_stem_planes_mfma", "k_pw_mfma<1, 4, 1>", "k_conv3x3_rows_mfma<1, 1>"):
        assert k in inst and inst[k]["kernel"]["MFMA"] > 0, k
        # (k_rfb_tail's channel loops are fully unrolled since round 5: no loop of it holds an MFMA)
        assert k == "k_rfb_tail" or inst[k]["hot_loop"].get("MFMA", 0) > 0, k
    h = inst["k_dwpw2_mfma<16, 1, true>"]["hot_loop"]
    assert h["MFMA"] == 64 and h["DPP"] >= 100 and h.get("cndmask", 0) <= 16, h  # (round 4's build: 135 selects in this loop)
