"""Row N4: the multi-stream batching scheduler of the C ABI (csrc/sched.cpp) -- the batching rule on the
CPU, and on the GPU streams of both variants and both output kinds through one scheduler, every frame checked against
the CPU oracle (router.rs:64-71 drop rule, inferer.rs:23,29-50)."""
import os
import time

import numpy as np
import pytest

from infercam_onnx_amd import scheduler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- the batching rule (no GPU) ----
def test_plan_round_robin_one_frame_per_stream_per_pass():
    # a busy stream cannot starve the others: everyone gets one frame before anyone gets a second
    assert scheduler.plan([30, 1, 1, 1], last=3, max_batch=4) == [1, 1, 1, 1]
    assert scheduler.plan([30, 1, 1, 1], last=3, max_batch=8) == [5, 1, 1, 1]
    assert scheduler.plan([30, 0, 2, 0], last=0, max_batch=5) == [3, 0, 2, 0]


def test_plan_starts_after_the_stream_served_last():
    # three streams, room for two: who is served rotates with `last`
    assert scheduler.plan([4, 4, 4], last=2, max_batch=2) == [1, 1, 0]
    assert scheduler.plan([4, 4, 4], last=0, max_batch=2) == [0, 1, 1]
    assert scheduler.plan([4, 4, 4], last=1, max_batch=2) == [1, 0, 1]
    assert scheduler.plan([5, 1, 0, 3], last=1, max_batch=4) == [1, 1, 0, 2]


def test_plan_takes_everything_when_it_fits_and_never_more_than_queued():
    assert scheduler.plan([5, 1, 0, 3], last=3, max_batch=32) == [5, 1, 0, 3]
    assert scheduler.plan([0, 0], last=0, max_batch=8) == [0, 0]
    assert scheduler.plan([40], last=0, max_batch=32) == [32]
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = int(rng.integers(1, 9))
        q = [int(v) for v in rng.integers(0, 12, n)]
        mb = int(rng.integers(1, 40))
        t = scheduler.plan(q, int(rng.integers(0, n)), mb)
        assert sum(t) == min(sum(q), mb) and all(a <= b for a, b in zip(t, q))
        # fairness: a stream with frames left over was given at least as many as any other minus one
        for i in range(n):
            if t[i] < q[i]:
                assert all(t[i] >= t[j] - 1 for j in range(n))


# ---- on the GPU ----
gpu = pytest.mark.gpu


def _models(weights, batch=8):
    from infercam_onnx_amd import nn, synth

    m320 = nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=weights, priors=synth.gen_priors(320, 240),
                             max_batch=batch, max_src=(640, 480), det_cap=256)
    m640 = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=weights, priors=synth.gen_priors(640, 480),
                             max_batch=batch, max_src=(640, 480), det_cap=256)
    return m320, m640


@gpu
def test_streams_of_both_variants_and_output_kinds(weights):
    """Five cameras on one scheduler: two detection-only and one annotated stream on UltraFace-320, one of each on
    UltraFace-640 (the reference fixes 320 for all, inferer.rs:23).  Every delivered frame equals the oracle's result
    for that stream's model; per stream, results arrive in push order; nothing is lost."""
    import oracle
    from helpers import assert_dets_match, dets_array
    from infercam_onnx_amd import synth

    m320, m640 = _models(weights)
    pri = {320: synth.gen_priors(320, 240), 640: synth.gen_priors(640, 480)}
    size = {320: (320, 240), 640: (640, 480)}
    try:
        with scheduler.Scheduler(m320, m640, ring_slots=16, max_wait_us=1000, det_cap=17640) as sch:  # (cap = K: nothing truncated)
            streams = {}
            for sid, (variant, annot) in enumerate([(320, False), (320, False), (320, True), (640, False), (640, True)]):
                streams[sid] = (sch.add_stream(1000 + sid, variant, annotate=annot, label_size=(1280, 720)), variant, annot)
            frames = {}
            for k in range(12):
                for sid, (idx, variant, annot) in streams.items():
                    w, h = size[variant]
                    j = synth.encode_jpeg(synth.synth_frame(40 + sid, k, w, h))
                    frames[(1000 + sid, k)] = j
                    while not sch.push(idx, j, tag=k):  # ring full: the router would drop; the test retries
                        time.sleep(0.0005)
            sch.flush()
            st = sch.stats()
            res = list(sch.results)
        assert st["delivered"] == len(frames) == len(res) and st["pushed"] - st["dropped"] == st["delivered"]
        assert st["frames_in_batches"] == st["delivered"] and st["batches"] == st["sent_full"] + st["sent_deadline"] + st["sent_idle"]
        by_stream = {}
        for r in res:
            by_stream.setdefault(r["stream_id"], []).append(r["tag"])
        assert all(tags == sorted(tags) and len(tags) == 12 for tags in by_stream.values())
        for r in res:
            sid = r["stream_id"] - 1000
            _, variant, annot = streams[sid]
            w, h = size[variant]
            j = frames[(r["stream_id"], r["tag"])]
            assert r["status"] == 0 and r["variant"] == variant
            assert_dets_match(dets_array(r["dets"]), oracle.infer_jpeg(j, w, h, weights, pri[variant]), what="sched")
            if annot:
                frame = oracle.draw_labels(oracle.jpeg_decode_rgb(j), dets_array(r["dets"]), 1280, 720)
                assert r["jpeg"] == oracle.jpeg_encode_rgb(frame, 95)
            else:
                assert r["jpeg"] is None
    finally:
        m320.close()
        m640.close()


@gpu
def test_full_ring_drops_the_new_frame_and_lone_frames_do_not_wait(weights):
    """router.rs:65: a frame that finds no free slot is dropped (push reports it); what was accepted is delivered in
    order.  A single frame pushed to an idle scheduler leaves at once in a batch of one."""
    from infercam_onnx_amd import nn, synth

    m320, m640 = _models(weights, batch=4)
    try:
        got = []
        with scheduler.Scheduler(m320, None, on_result=got.append, ring_slots=2, max_wait_us=500, max_inflight=2) as sch:
            s0 = sch.add_stream(7, 320)
            j = synth.encode_jpeg(synth.synth_frame(3, 3, 320, 240))
            assert sch.push(s0, j, tag=99)
            sch.flush()
            assert [r["tag"] for r in got] == [99] and got[0]["batch_fill"] == 1 and sch.stats()["sent_idle"] == 1
            assert got[0]["total_ms"] < 50
            got.clear()
            accepted = [k for k in range(300) if sch.push(s0, j, tag=k)]
            sch.flush()
            st = sch.stats()
            assert [r["tag"] for r in got] == accepted
            assert st["dropped"] == 300 - len(accepted) and st["delivered"] == 1 + len(accepted)
            corrupt = j[:150] + bytes(40) + j[190:len(j) // 3]
            got.clear()
            assert sch.push(s0, corrupt, tag=1) and sch.push(s0, j, tag=2)
            sch.flush()
            assert [(r["tag"], r["status"]) for r in got] == [(1, nn.UFD_E_DECODE), (2, 0)] and got[0]["dets"] is None
            with pytest.raises(nn.UfdError):
                sch.add_stream(8, 640)  # no 640 handle was given to this scheduler
    finally:
        m320.close()
        m640.close()
