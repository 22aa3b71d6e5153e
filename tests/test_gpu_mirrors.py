"""The host-side mirrors of the reference interface on the GPU: the C++ `nn.hpp` wrapper (a port of
infer_server/tests/integration_tests.rs) and the Python `Inferer` loop (inferer.rs:29-50)."""
import os
import queue
import subprocess
import sys

import numpy as np
import pytest

from helpers import assert_dets_match, dets_array

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_integration_test(tmp_path, weights):
    exe = str(tmp_path / "integration_test")
    lib_dir = os.path.join(ROOT, "infercam_onnx_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "integration_test.cpp"),
                           "-o", exe, "-L" + lib_dir, "-lufacehip", "-Wl,-rpath," + lib_dir])
    wfile = str(tmp_path / "w.f32")
    np.asarray(weights, np.float32).tofile(wfile)
    # (output to a file, unbuffered: if the process ever hangs, the assertion shows how far it got)
    log = str(tmp_path / "out.txt")
    with open(log, "w") as f:
        try:
            rc = subprocess.run(["stdbuf", "-o0", exe, os.path.join(ROOT, "tests", "golden", "test_pics"), wfile], stdout=f,
                                stderr=subprocess.STDOUT, timeout=300).returncode
        except subprocess.TimeoutExpired:
            rc = "timeout"
    text = open(log).read()
    assert rc == 0, "rc %s\n%s" % (rc, text)
    assert text.strip().endswith("ok") and text.count("faces=") == 8


def _real_model_path(variant):
    return os.path.join(os.environ.get("XDG_CACHE_HOME", os.path.expanduser("~/.cache")), "infercam_onnx",
                        "ultraface-RFB-%d.onnx" % variant)


def test_reference_face_counts_if_real_model_present(oracle_lib):
    """integration_tests.rs:20-35 proper: only runs when the real ONNX file is in the cache path."""
    import json
    from infercam_onnx_amd import nn
    from helpers import REFERENCE_PINS

    path = _real_model_path(640)
    if not os.path.exists(path):
        REFERENCE_PINS["face_counts_640"] = "NOT CHECKED (%s absent)" % path
        pytest.skip("real UltraFace weights not available offline (reference downloads them, nn.rs:155-162)")
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "test_pics.json")))
    with nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights_path=path, max_src=(1280, 1024)) as m:
        for f, info in meta.items():
            jpeg = open(os.path.join(ROOT, "tests", "golden", "test_pics", f), "rb").read()
            assert len(m.infer_jpeg(jpeg)) == info["reference_face_count"], f
    REFERENCE_PINS["face_counts_640"] = "checked: 8 pictures"


def test_real_320_model_loads_and_agrees_with_the_oracle_if_present(oracle_lib):
    """The server's own operating point (inferer.rs:23: W320H240, 0.5, 0.5) on the zoo file version-RFB-320.onnx: the
    reference pins no result for it, so when the file is there the GPU path is compared with the oracle running the
    SAME file's weights on the reference's 8 pictures.  Skipped (and reported) when the file is absent."""
    import json
    from infercam_onnx_amd import nn
    from helpers import REFERENCE_PINS

    path = _real_model_path(320)
    if not os.path.exists(path):
        REFERENCE_PINS["zoo_weights_320"] = "NOT CHECKED (%s absent)" % path
        pytest.skip("real UltraFace-320 weights not available offline (nn.rs:21-22)")
    from infercam_onnx_amd import synth

    w, pri = nn.load_onnx(path, 320)
    pri = pri if pri is not None else synth.gen_priors(320, 240)
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "test_pics.json")))
    with nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights_path=path, max_src=(1280, 1024)) as m:
        for f in meta:
            jpeg = open(os.path.join(ROOT, "tests", "golden", "test_pics", f), "rb").read()
            ref = oracle_lib.infer_jpeg(jpeg, 320, 240, w, pri, 0.5, 0.5)
            assert_dets_match(dets_array(m.infer_jpeg(jpeg)), ref, what=f)
    REFERENCE_PINS["zoo_weights_320"] = "checked against the oracle on 8 pictures"


def test_inferer_loop(oracle_lib, weights):
    from infercam_onnx_amd import nn, synth
    from infercam_onnx_amd.inferer import Inferer

    W, H = 320, 240
    pri = synth.gen_priors(W, H)
    model = nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=weights, priors=pri, max_batch=4,
                              max_src=(1280, 720), det_cap=4420)
    rx = queue.Queue()
    got = {}
    jpegs = [synth.encode_jpeg(synth.synth_frame(61, i, 1280, 720)) for i in range(6)]  # router stamps 1280x720
    for i, j in enumerate(jpegs):
        rx.put((1280, 720, j if i != 3 else j[:777], lambda r, i=i: got.__setitem__(i, r)))
    rx.put(None)
    Inferer(rx, model=model, max_batch=4).run()
    assert sorted(got) == list(range(6))
    assert got[3] == (None, nn.UFD_E_DECODE)  # corrupt frame skipped, loop keeps going
    for i in (0, 1, 2, 4, 5):
        dets, st = got[i]
        assert st == 0
        ref = oracle_lib.infer_jpeg(jpegs[i], W, H, weights, pri, 0.5, 0.5)
        x = oracle_lib.normalize_nchw(oracle_lib.resize_triangle(oracle_lib.jpeg_decode_rgb(jpegs[i]), W, H))
        scores, _ = oracle_lib.forward(x, weights, pri)
        assert_dets_match(dets_array(dets), ref, scores=scores)
    model.close()


def test_inferer_loop_annotate(oracle_lib, weights):
    """The whole Inferer::run iteration (inferer.rs:35-46): the sender receives `as_jpeg_stream_item(&buf)` of the
    annotated frame; a frame that fails to decode sends nothing (inferer.rs:37, `if let Ok`)."""
    from infercam_onnx_amd import nn, synth
    from infercam_onnx_amd.inferer import Inferer

    W, H = 320, 240
    pri = synth.gen_priors(W, H)
    model = nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=weights, priors=pri, max_batch=4,
                              max_src=(1280, 720), det_cap=4420)
    rx = queue.Queue()
    got = {}
    jpegs = [synth.encode_jpeg(synth.synth_frame(67, i, 640, 480)) for i in range(5)]
    labels = [(1280, 720), (1280, 720), (1280, 720), (640, 480), (1280, 720)]  # a second label size splits the batch
    for i, j in enumerate(jpegs):
        rx.put((labels[i][0], labels[i][1], j if i != 1 else j[:555], lambda r, i=i: got.__setitem__(i, r)))
    rx.put(None)
    Inferer(rx, model=model, max_batch=4, annotate=True).run()
    assert sorted(got) == [0, 2, 3, 4]
    for i in (0, 2, 3, 4):
        # drawn with the detections the GPU reports (compared with the oracle's on their own: a corner may sit on a
        # rounding boundary), then the oracle's draw + encode + multipart framing
        dets = model.infer_jpeg(jpegs[i])
        assert_dets_match(dets_array(dets), oracle_lib.infer_jpeg(jpegs[i], W, H, weights, pri, 0.5, 0.5), what="inferer")
        frame = oracle_lib.draw_labels(oracle_lib.jpeg_decode_rgb(jpegs[i]), dets_array(dets), *labels[i])
        assert got[i] == oracle_lib.stream_item(oracle_lib.jpeg_encode_rgb(frame, 95)), i
    model.close()


@pytest.mark.gpu
def test_whole_file_fuzz_never_faults():
    """tools/fuzz_gpu.py: random corruptions over whole JPEG files (headers, markers, entropy data)
    in mixed batches, through the detection path and the annotate + re-encode path.  A GPU memory fault aborts the child process, so it runs as one."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # thumbnails through UltraFace-320; 640x480-class frames through UltraFace-640; all frames exactly 640x480 (round 5: the
    # 4:2:0 and 4:2:2 ones take the fused stem in the same damaged batch)
    # ... and (round 6) the libjpeg-turbo-written layout fixtures as seeds: damaged 4:1:1 / 4:1:0 (ten blocks per MCU) / 4:4:0 / RGB /
    # non-interleaved streams meet the device entropy decoder and the generic upsampler
    for extra in (["40"], ["30", "big"], ["30", "model"], ["60", "layouts"]):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gpu.py")] + extra, cwd=root, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "fuzz ok" in r.stdout


@pytest.mark.gpu
def test_concurrent_callers_on_one_handle(weights):
    """include/ufd.h "Threading": any thread may call a handle; calls serialise inside.  Four threads
    mix synchronous batches, single frames and RGB calls on ONE handle and all get the results a
    single-threaded run gives."""
    import threading
    from infercam_onnx_amd import nn, synth

    m = nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=weights, priors=synth.gen_priors(320, 240),
                          max_batch=4, max_src=(640, 480), det_cap=4420)
    try:
        frames = [synth.synth_frame(95, i, 320, 240) for i in range(4)]
        jpegs = [synth.encode_jpeg(f, restart_rows=(i % 2)) for i, f in enumerate(frames)]
        ref_batch, _ = m.infer_jpeg_batch(jpegs)
        ref_single = [m.infer_jpeg(j) for j in jpegs]  # (batch 1 picks other kernels: fp32 rounding apart from batch 4)
        ref_rgb = m.run(frames[0])
        errors = []

        def worker(k):
            try:
                for it in range(12):
                    if (k + it) % 3 == 0:
                        got, st = m.infer_jpeg_batch(jpegs)
                        assert st == [0] * 4 and got == ref_batch
                    elif (k + it) % 3 == 1:
                        assert m.infer_jpeg(jpegs[k % 4]) == ref_single[k % 4]
                    else:
                        assert m.run(frames[0]) == ref_rgb
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
    finally:
        m.close()


@pytest.mark.parametrize("args", [["16", "clean"], ["8", "small"], ["8", "hostentropy"]])
def test_soak_host_memory_does_not_grow_per_batch(args):
    """tools/soak.py, 16 s of the bench's submit / wait loop on the host-bytes path (~25 000 batches): the process's resident
    set must not grow with the batch count.  Round 4 found ROCm 7.2's runtime keeping ~2 KB of host memory per event recorded
    directly behind an asynchronous copy (12 GB per hour at the bench's rate); the library now puts an empty kernel between
    the two (model.cpp: record_behind_copy; tools/ubench/leak_probe2.hip shows the runtime's behaviour on its own).
    "small": 8 s of two-frame batches (~45 000, every 50th with a truncated frame and other Huffman tables) -- the forms a
    batch of a few frames takes: staging block in and results out by kernels on the context's stream, no copy-engine transfer.
    "hostentropy": the coefficient slabs of host-decoded batches are copied on the context's stream with an event behind them."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py")] + args, cwd=root, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1500:] + r.stderr[-500:]


def test_create_replicas_runs_the_rccl_broadcast(tmp_path, oracle_lib, weights):
    """ufd_create_replicas from a plain C++ host process (tests/cpp/replicas_test.cpp: no Python, no torch -- the
    environment of the reference's one Rust binary) on every GPU of this box: ncclCommInitAll + ncclBroadcast of the packed
    weight image and the priors really execute (a one-GPU box forms a one-rank communicator; with more GPUs every other
    handle's weights exist ONLY through the broadcast, since it is created from a zero blob), and every replica gives the
    oracle's detections.  (In-process after `import torch` this would mix torch's bundled ROCm runtime with the system one
    RCCL is loaded from, which RCCL's start-up does not survive: hence the separate process.)"""
    from infercam_onnx_amd import nn, synth

    W, H = 320, 240
    pri = synth.gen_priors(W, H)
    exe = str(tmp_path / "replicas_test")
    lib_dir = os.path.join(ROOT, "infercam_onnx_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "replicas_test.cpp"),
                           "-o", exe, "-L" + lib_dir, "-lufacehip", "-Wl,-rpath," + lib_dir])
    wfile, jfile = str(tmp_path / "w.f32"), str(tmp_path / "f.jpg")
    np.asarray(weights, np.float32).tofile(wfile)
    jpeg = synth.encode_jpeg(synth.synth_frame(synth.DEFAULT_FRAME_SEED, 2, W, H))
    open(jfile, "wb").write(jpeg)
    out = subprocess.run([exe, wfile, jfile, "8"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
    ref = oracle_lib.infer_jpeg(jpeg, W, H, weights, pri, 0.5, 0.5)
    assert len(ref) > 0
    lines = [l.split() for l in out.stdout.splitlines() if l.startswith("replica ")]
    assert len(lines) >= 1
    # ... and ONE scheduler over those replicas delivered every frame of 2 n + 1 streams from its stream's replica (checked
    # in the C++ process against the first replica's detections, which are checked against the oracle below)
    sched = [l.split() for l in out.stdout.splitlines() if l.startswith("sched streams")]
    assert len(sched) == 1 and int(sched[0][2]) == 2 * len(lines) + 1 and int(sched[0][4]) == 6 * (2 * len(lines) + 1), out.stdout
    for l in lines:
        cnt = int(l[11])
        got = np.array([float(v) for v in l[12:]], np.float32).reshape(cnt, 5)
        assert_dets_match(got, ref, what="replica " + l[1])
        assert l[5].count(":") == 2  # pci address
    # the n = 2 group of ncclBroadcasts (replicas.cpp) on this one-GPU box: device 0 listed twice under a test-only switch.
    # Either RCCL forms two ranks on one device and the second handle (created from a zero blob) detects what the first does,
    # or it refuses and the library hands the refusal back as a status with out[] cleared; which one is printed.
    dup = subprocess.run([exe, wfile, jfile, "8", "dup"], capture_output=True, text=True, timeout=300)
    assert dup.returncode == 0 and dup.stdout.strip().endswith("ok"), dup.stdout + dup.stderr
    print("replicas n=2 on one device:", dup.stdout.strip().splitlines()[0])
    from helpers import SESSION_NOTES

    SESSION_NOTES["ufd_create_replicas with n = 2 on one device"] = dup.stdout.strip().splitlines()[0][:160]
    # with a device present, ids out of range / listed twice are still refused (argument checks: tests/test_host_logic.py)
    for bad in ([len(lines)], [0, 0], [-1]):
        with pytest.raises(nn.UfdError) as e:
            nn.UltrafaceModel.create_replicas(nn.UltrafaceVariant.W320H240, 0.5, 0.5, bad, weights=weights, priors=pri)
        assert e.value.code == nn.UFD_E_ARG


def test_numa_pinning_follows_the_gpu_and_can_be_switched_off(weights):
    """Host placement: the handle reports its PCI address and NUMA node; when the node is known its host threads are
    pinned to that node's CPUs inside the process's affinity mask (never to a CPU outside it); UFD_FLAG_NO_NUMA_PIN
    pins nothing.  Results do not depend on it."""
    from infercam_onnx_amd import nn, synth

    pri = synth.gen_priors(320, 240)
    jpeg = synth.encode_jpeg(synth.synth_frame(3, 3, 320, 240))
    allowed = os.sched_getaffinity(0)
    res = []
    for off in (False, True):
        with nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=weights, priors=pri, no_numa_pin=off) as m:
            pl = m.placement()
            res.append(dets_array(m.infer_jpeg(jpeg)))
            assert pl["pci"].count(":") == 2, pl
            if off or pl["numa_node"] < 0:
                assert pl["pinned_cpus"] == 0 and pl["cpu_list"] == ""
            elif pl["pinned_cpus"]:  # (0: the node's CPUs and this process's affinity mask do not intersect)
                cpus = set()
                for part in pl["cpu_list"].split(","):
                    a, _, b = part.partition("-")
                    cpus |= set(range(int(a), int(b or a) + 1))
                assert len(cpus) == pl["pinned_cpus"] and cpus <= allowed
                node_cpus = open("/sys/devices/system/node/node%d/cpulist" % pl["numa_node"]).read().strip()
                assert node_cpus  # the node the kernel reports for the GPU exists
    assert np.array_equal(res[0], res[1])
    assert os.sched_getaffinity(0) == allowed  # the CALLER's thread is never re-pinned
