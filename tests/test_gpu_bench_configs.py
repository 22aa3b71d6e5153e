"""GPU parity at the configurations bench.py measures (BASELINE.json configs C1, C3, C5), through
the exact submission paths the bench uses: staged (HBM-resident) and host-bytes batches, six in
flight over the handle's device contexts, at the bench's batch sizes, on the bench's own
frame pool (which holds frames with > 256 and > 2048 NMS candidates).  Every frame's detection
list is compared with the CPU oracle (inferer.rs:35-37 + nn.rs:178-186)."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import assert_dets_match, dets_array, dets_from_ctypes, oracle_many

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _model(variant, weights, **kw):
    from infercam_onnx_amd import nn, synth

    v = nn.UltrafaceVariant.W640H480 if variant == 640 else nn.UltrafaceVariant.W320H240
    W, H = v.width_height()
    return nn.UltrafaceModel(v, 0.5, 0.5, weights=weights, priors=synth.gen_priors(W, H), **kw)


def _pipeline(model, batches, submit, rounds, depth):
    """bench.py's run_steps: `depth` batches in flight, batches cycled; yields (batch index, dets per frame)."""
    inflight, out = [], []

    def drain():
        bi, t = inflight.pop(0)
        b = model._pending[t]
        model.wait(t, collect=False)
        assert all(st in (0, -4) for st in b.status), list(b.status)  # -4: more detections than det_cap
        out.append((bi, [dets_from_ctypes(b.out, b.cnt, model.det_cap, i) for i in range(b.count)], [int(c) for c in b.cnt]))

    for s in range(rounds * len(batches)):
        if len(inflight) >= depth:
            drain()
        bi = s % len(batches)
        inflight.append((bi, submit(batches[bi])))
    while inflight:
        drain()
    return out


def _check_pool(model, jpegs, B, refs, depth=6, rounds=2):
    from infercam_onnx_amd import nn

    nb = len(jpegs) // B
    cap = model.det_cap
    excused = 0
    for mode in ("staged", "host"):
        if mode == "staged":
            batches = [model.stage_jpeg_batch(jpegs[i * B:(i + 1) * B]) for i in range(nb)]
            results = _pipeline(model, batches, model.submit_staged, rounds, min(depth, nb))
        else:
            # one set of output arrays per batch in flight: each batch object is in flight once at a time
            batches = [model._prep_batch(jpegs[i * B:(i + 1) * B]) for i in range(nb)]
            results = _pipeline(model, batches, model.submit_jpeg_batch, rounds, min(depth, nb))
        assert len(results) == rounds * nb
        for bi, dets, counts in results:
            for i, (d, n) in enumerate(zip(dets, counts)):
                ref = refs[bi * B + i]
                assert n >= len(d)
                if n > cap:  # truncated to cap: the first `cap` detections must be the oracle's first `cap`
                    ref = ref[:cap]
                excused += assert_dets_match(d, ref, what="%s batch %d frame %d" % (mode, bi, i))
        if mode == "staged":
            for b in batches:
                model.free_staged(b)
    return excused


@pytest.fixture(scope="module")
def pool_c3():
    from infercam_onnx_amd import synth

    return synth.synth_jpeg_pool(0, 256, 640, 480, quality=90, subsampling="4:2:0")


@pytest.fixture(scope="module")
def refs_c3(pool_c3, oracle_lib, weights):
    from infercam_onnx_amd import synth

    pri = synth.gen_priors(640, 480)
    return oracle_many(lambda j: oracle_lib.infer_jpeg(j, 640, 480, weights, pri, 0.5, 0.5), pool_c3)


def test_c3_batch32_bench_pipeline_matches_oracle(pool_c3, refs_c3, weights):
    """BASELINE C3 exactly as bench.py runs it: UltraFace-640, 640x480 stream, batch 32, the 256-frame
    bench pool, staged + host-bytes submission, 6 batches in flight over the handle's contexts; detections of
    every frame (1 ... > 2000 NMS candidates) against the oracle.  det_cap 512 also exercises the
    tail copy of frames with more than 256 detections under pipelining."""
    cands = sorted(len(r) for r in refs_c3)
    m = _model(640, weights, max_batch=32, max_src=(640, 480), det_cap=512)
    try:
        excused = _check_pool(m, pool_c3, 32, refs_c3)
    finally:
        m.close()
    print("C3: detections per frame min/median/max = %d/%d/%d; excused borderline detections: %d" %
          (cands[0], cands[len(cands) // 2], cands[-1], excused))


def test_c3_batch32_bench_handle_det_cap_256(pool_c3, refs_c3, weights):
    """The bench's own handle parameters (det_cap=256): truncated frames report the true count and
    the oracle's first 256 detections."""
    m = _model(640, weights, max_batch=32, max_src=(640, 480), det_cap=256, profile=True)
    try:
        _check_pool(m, pool_c3[:128], 32, refs_c3[:128], rounds=3)
    finally:
        m.close()


def test_c3_422_mjpg_stream_batch32_matches_oracle(oracle_lib, weights):
    """The second synthetic input SURVEY 8(d) names, as `bench.py --subsampling 4:2:2 --no-dht --restart-rows 1` runs it:
    UltraFace-640, a 640x480 4:2:2 stream WITHOUT DHT segments and with one restart interval per MCU row (what a UVC camera's
    MJPG looks like: cam_sender/src/sensors.rs:18-68), batch 32, six batches in flight, staged + host-bytes submission; the
    fused stem (round 5: h2v1 per frame) is the path taken; detections of every frame against the oracle."""
    from infercam_onnx_amd import synth

    pool = [synth.strip_dht(j) for j in synth.synth_jpeg_pool(0, 128, 640, 480, quality=90, subsampling="4:2:2", restart_rows=1)]
    pri = synth.gen_priors(640, 480)
    refs = oracle_many(lambda j: oracle_lib.infer_jpeg(j, 640, 480, weights, pri, 0.5, 0.5), pool)
    m = _model(640, weights, max_batch=32, max_src=(640, 480), det_cap=512, profile=True)
    try:
        excused = _check_pool(m, pool, 32, refs)
        names = {p["name"] for p in m.profile_read() if p["launches"]}
        assert any(n.startswith("stem_planes_mfma:") for n in names) and not any(n.startswith("upsample_norm") for n in names), names
        assert "huff_write" in names  # the device entropy decoder took the table-less streams
    finally:
        m.close()
    print("C3 4:2:2 MJPG: excused borderline detections: %d" % excused)


def test_host_stats_account_for_every_batch(pool_c3, weights):
    """ufd_host_stats (bench.py's `host` object): 24 batches through the bench's submit / wait loop are all counted, each
    with its launches, a device-time span on its context and (between consecutive batches of a context) a gap; a reset
    zeroes the sums."""
    m = _model(640, weights, max_batch=32, max_src=(640, 480), det_cap=256)
    try:
        batches = [m._prep_batch(pool_c3[i * 32:(i + 1) * 32]) for i in range(6)]
        _pipeline(m, batches, m.submit_jpeg_batch, 1, 6)  # warm
        m.host_stats_reset()
        _pipeline(m, batches, m.submit_jpeg_batch, 4, 6)
        hs = m.host_stats()
        assert hs["batches"] == 24, hs
        assert 15 <= hs["launches_per_batch"] <= 80, hs
        pb = hs["per_batch_us"]
        assert pb["header_scan"] > 0 and pb["staging_memcpy"] > 0 and pb["launch_issue"] > 0 and pb["wait"] >= 0, hs
        assert len(hs["worker_busy_share"]) == len(hs["gpu_span_share"]) >= 2, hs
        assert all(0 < v < 1.0 for v in hs["worker_busy_share"]), hs
        assert all(0 < v <= 1.02 for v in hs["gpu_span_share"]), hs
        assert all(0.05 < v < 50 for v in hs["gpu_span_ms_per_batch"]), hs
        assert all(v >= 0 for v in hs["gpu_idle_gap_us_per_batch"]), hs
        m.host_stats_reset()
        z = m.host_stats()
        assert z["batches"] == 0 and z["per_batch_us"]["launch_issue"] == 0, z
        print("host stats:", hs)
    finally:
        m.close()


def test_c5_batch16_1280x720_matches_oracle(oracle_lib, weights):
    """BASELINE C5: 1280x720 frames -> UltraFace-640 (Triangle resize 2.0 x 1.5 on the GPU), batch 16."""
    from infercam_onnx_amd import synth

    pool = synth.synth_jpeg_pool(0, 96, 1280, 720, quality=90, subsampling="4:2:0")
    pri = synth.gen_priors(640, 480)
    refs = oracle_many(lambda j: oracle_lib.infer_jpeg(j, 640, 480, weights, pri, 0.5, 0.5), pool)
    m = _model(640, weights, max_batch=16, max_src=(1280, 720), det_cap=512)
    try:
        _check_pool(m, pool, 16, refs)
    finally:
        m.close()


def test_c2_batch1_320_stream_matches_oracle(oracle_lib, weights):
    """BASELINE C2: UltraFace-320 on a 320x240 stream, batch 1, one at a time (the latency form)."""
    from infercam_onnx_amd import synth

    pool = synth.synth_jpeg_pool(0, 24, 320, 240)
    pri = synth.gen_priors(320, 240)
    m = _model(320, weights, max_batch=1, max_src=(320, 240), det_cap=4420)
    try:
        for j in pool:
            got = dets_array(m.infer_jpeg(j))
            assert_dets_match(got, oracle_lib.infer_jpeg(j, 320, 240, weights, pri, 0.5, 0.5), what="C2")
    finally:
        m.close()


def test_production_plan_tensors_at_batch32(pool_c3, oracle_lib, weights):
    """The plan the product issues at the bench's batch size (chained m1->m2 / m3->m4, stacked RFB
    reduce convs, merged dilated launch, summed RFB tail, non-split-K kernel variants): every tensor
    that exists in that plan -- m0, m2.pw, m4.pw ... the RFB output, the 8 head maps -- against the
    oracle's layer outputs, <= 1e-5 relative, for all 32 frames."""
    from infercam_onnx_amd import nn, synth

    W, H = 640, 480
    pri = synth.gen_priors(W, H)
    x = np.stack(oracle_many(lambda j: oracle_lib.normalize_nchw(oracle_lib.jpeg_decode_rgb(j)), pool_c3[:32]))
    m = _model(640, weights, max_batch=32, max_src=(640, 480), tap_layers=True)
    try:
        scores, boxes = m.debug_forward(x)
        present, absent = [], []
        for f in range(32):
            rs, rb, outs = oracle_lib.forward(x[f], weights, pri, layers=True)
            for li, ref in enumerate(outs):
                try:
                    got = m.debug_layer_output(li, f).reshape(ref.shape)
                except nn.UfdError as e:
                    assert e.code == nn.UFD_E_STATE
                    if f == 0:
                        absent.append(li)
                    continue
                if f == 0:
                    present.append(li)
                scale = max(np.abs(ref).max(), 1e-6)
                err = np.abs(got - ref).max() / scale
                assert err <= 1e-5, "layer %d frame %d: rel err %g" % (li, f, err)
            assert np.abs(scores[f] - rs).max() <= 1e-5
            assert np.abs(boxes[f] - rb).max() <= 1e-5
        # the taps VERDICT r1 names must be among the tensors that exist
        for li in (0, 4, 8, 24, 26, 28, 36, 38, 44, 46, 50, 51):
            assert li in present, (li, present)
        assert 2 in absent and 6 in absent and 23 in absent  # m1.pw, m3.pw (chained), rfb.linear (summed)
    finally:
        m.close()


@pytest.mark.parametrize("variant,batch", [(640, 32), (320, 5)])
def test_bn_folded_like_weight_ranges_through_the_product_plan(pool_c3, oracle_lib, variant, batch):
    """All other CNN parity runs on He-normal weights with one scale per layer.  A BatchNorm-folded checkpoint -- what the
    reference's zoo file is (nn.rs:21-22, not available offline) -- has a scale of its own per OUTPUT CHANNEL and biases
    tens of times larger: `synth.bn_folded_like_weights` (per-channel factors log-uniform in [1/4, 4], biases N(0, 0.5^2)).
    Every tensor of the product plan against the oracle, <= 1e-5 of the tensor's largest value, scores <= 1e-5, boxes
    <= 1e-5 relative -- at the bench's batch size (non-split-K instances) and at a small batch (split-K instances)."""
    from infercam_onnx_amd import nn, synth

    W, H = (640, 480) if variant == 640 else (320, 240)
    weights = synth.bn_folded_like_weights()
    pri = synth.gen_priors(W, H)
    jpegs = pool_c3[:batch] if variant == 640 else synth.synth_jpeg_pool(3, batch, W, H, quality=90, subsampling="4:2:0")
    x = np.stack(oracle_many(lambda j: oracle_lib.normalize_nchw(oracle_lib.jpeg_decode_rgb(j)), jpegs))
    m = _model(variant, weights, max_batch=batch, max_src=(W, H), tap_layers=True)
    try:
        scores, boxes = m.debug_forward(x)
        checked = 0
        for f in range(0, batch, max(1, batch // 4)):  # (the oracle's forward is ~1 s per 640 frame)
            rs, rb, outs = oracle_lib.forward(x[f], weights, pri, layers=True)
            for li, ref in enumerate(outs):
                try:
                    got = m.debug_layer_output(li, f).reshape(ref.shape)
                except nn.UfdError as e:
                    assert e.code == nn.UFD_E_STATE  # (a tensor that does not exist in the fused plan)
                    continue
                scale = max(np.abs(ref).max(), 1e-6)
                assert np.isfinite(ref).all() and np.abs(got - ref).max() / scale <= 1e-5, (li, f, np.abs(got - ref).max() / scale)
                checked += 1
            assert np.abs(scores[f] - rs).max() <= 1e-5
            assert (np.abs(boxes[f] - rb) <= 1e-5 * np.maximum(1.0, np.abs(rb))).all()
        assert checked >= 4 * 28, checked  # (29 tensors exist in the 640 plan since k_rfb_tail took the three dilated RFB convs inside its launch)
        w10, b10 = synth.layer_params(weights, 10)  # the generator's point: one backbone layer's channels differ by > 8x
        norms = np.sqrt((w10.reshape(w10.shape[0], -1) ** 2).sum(1))
        assert norms.max() / norms.min() > 8.0 and np.abs(b10).max() > 0.5, (norms.max() / norms.min(), np.abs(b10).max())
    finally:
        m.close()


# ---------------------------------------------------------------- C1: the reference's own pictures
def _pics():
    meta = json.load(open(os.path.join(G, "test_pics.json")))
    return sorted(meta.items())


@pytest.mark.parametrize("name,info", _pics())
def test_reference_pictures_decode_on_gpu_to_libjpeg_turbo_pixels(weights, name, info):
    """integration_tests.rs:20-31's eight pictures (progressive 4:2:0): the GPU decode equals the
    libjpeg-turbo decode whose sha256 is committed in tests/golden/test_pics.json."""
    jpeg = open(os.path.join(G, "test_pics", name), "rb").read()
    m = _model(640, weights, max_batch=1, max_src=(1280, 1024))
    try:
        rgb = m.debug_decode_jpeg(jpeg)
    finally:
        m.close()
    assert list(rgb.shape) == info["shape"]
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == info["sha256"]


@pytest.mark.parametrize("variant", [320, 640])
def test_reference_pictures_infer_matches_oracle(oracle_lib, weights, variant):
    """C1's workload on the GPU: each test picture (640 x {427...960}, stretched to the model
    size) through decode -> resize -> UltraFace-{320,640} -> NMS, against the oracle."""
    from infercam_onnx_amd import synth

    W, H = (640, 480) if variant == 640 else (320, 240)
    pri = synth.gen_priors(W, H)
    m = _model(variant, weights, max_batch=8, max_src=(1280, 1024), det_cap=17640)
    try:
        jpegs = [open(os.path.join(G, "test_pics", n), "rb").read() for n, _ in _pics()]
        refs = oracle_many(lambda j: oracle_lib.infer_jpeg(j, W, H, weights, pri, 0.5, 0.5), jpegs, threads=8)
        res, status = m.infer_jpeg_batch(jpegs)  # mixed sizes in one batch
        assert status == [0] * 8
        for (n, _), r, ref, j in zip(_pics(), res, refs, jpegs):
            assert_dets_match(dets_array(r), ref, what="%s @%d" % (n, variant))
            assert_dets_match(dets_array(m.infer_jpeg(j)), ref, what="%s @%d single" % (n, variant))
    finally:
        m.close()


def test_bench_script_single_and_two_rank_rehearsal():
    """bench.py's own contract on the box: the N=1 line carries every field the driver reads (roofline, cpu_baseline,
    verified), and the N>1 script path under torch.distributed.run (C4's launch: one process per rank, weight broadcast,
    barrier-bracketed timing, max over ranks, per-rank records) runs end to end.  The box has one GPU: the two ranks
    share cuda:0 and exchange over gloo (`--rehearse-one-gpu`), so the figure is not a measurement -- RCCL itself and
    distinct devices are the driver's 8-GPU run."""
    import socket
    import subprocess
    import sys

    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--cpu-seconds", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "verified"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["value"] > 1000 and line["verified"]["max_abs_err"] < 1e-3
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--rehearse-one-gpu"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "rehearsal" in line["config"]
    assert len(line["config"]["ranks"]) == 2 and line["config"]["global_batch"] == 64
    assert "cpu_baseline" not in line  # N = 1 only
    assert "host" in json.loads(r.stdout.strip().splitlines()[-1])  # every rank-0 line carries the host object

    # C4's other form: ONE process, one scheduler over the replicas (two handles on cuda:0 here)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--one-process", "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--pool", "64", "--rehearse-one-gpu"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["stream_to_replica"] == [0, 1] and line["config"]["replica_frames"] == [192, 192]
    assert line["value"] > 1000 and line["detections_per_frame"] > 0 and "rehearsal" in line["config"]


def test_rccl_backend_collectives_coexist_with_the_library():
    """The leg of the N > 1 path the two-rank rehearsal above cannot reach (it exchanges over gloo): torch.distributed's
    "nccl" backend -- torch's bundled RCCL on torch's bundled HIP runtime -- with device tensors, in ONE process with
    libufacehip.so (linked against the system ROCm runtime).  tools/nccl_coexist_probe.py runs bench.py's sequence at world
    size 1 on the box's GPU: init_process_group("nccl", device_id=...), the weight broadcast, a handle created and a
    batch run, then barrier / all_reduce(MAX) / all_gather.  (The reverse mix does not work -- the system librccl
    dlopen'ed into a process that has torch's runtime loaded fails in ncclCommInitAll -- which is why
    ufd_create_replicas is tested from a C++ host: test_gpu_mirrors.py.)  More than one GPU stays the driver's run."""
    import socket
    import subprocess
    import sys

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_coexist_probe.py")], cwd=ROOT, capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "nccl coexist ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
