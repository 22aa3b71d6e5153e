"""Ingest boundary: wire format + router rule restated from the reference
(common/src/protocol.rs, data_socket.rs:38, socket_sender.rs:68-90, router.rs:56-72)."""
import queue
import struct

import numpy as np
import pytest

from infercam_onnx_amd import ingest


def test_bincode_frame_msg_known_bytes_and_roundtrip():
    """The reference's own protocol test (protocol.rs:37-49) uses FrameMsg{id: "bla", data: [1,2,3]};
    bincode 1.3.3 default options: u32-LE variant index, u64-LE lengths."""
    wire = ingest.encode_frame_msg("bla", bytes([1, 2, 3]))
    assert wire == bytes.fromhex("01000000" "0300000000000000" "626c61" "0300000000000000" "010203")
    assert ingest.decode_proto_msg(wire) == ("frame", "bla", bytes([1, 2, 3]))
    assert ingest.decode_proto_msg(wire + b"trailing") == ("frame", "bla", bytes([1, 2, 3]))  # bincode::deserialize allows it
    req = ingest.encode_connect_req("cam0")
    assert req == bytes.fromhex("00000000" "0400000000000000") + b"cam0"
    assert ingest.decode_proto_msg(req) == ("connect", "cam0")
    big = bytes(np.random.default_rng(0).integers(0, 256, 70000, dtype=np.uint8))
    assert ingest.decode_proto_msg(ingest.encode_frame_msg("Ünï", big)) == ("frame", "Ünï", big)
    assert ingest.decode_proto_msg(ingest.encode_frame_msg("", b"")) == ("frame", "", b"")


@pytest.mark.parametrize("bad", [b"", b"\x01\x00\x00", struct.pack("<I", 2) + b"\0" * 16,
                                 struct.pack("<IQ", 1, 5) + b"ab",                       # id length beyond the message
                                 struct.pack("<IQ", 1, 1) + b"a" + struct.pack("<Q", 9) + b"x",  # data length beyond it
                                 struct.pack("<IQ", 1, 2) + b"\xff\xfe" + struct.pack("<Q", 0),  # id is not UTF-8
                                 struct.pack("<IQ", 0, 1 << 62)])
def test_bincode_errors(bad):
    with pytest.raises(ValueError):
        ingest.decode_proto_msg(bad)


def test_length_delimited_reader_any_split():
    rng = np.random.default_rng(3)
    payloads = [bytes(rng.integers(0, 256, int(n), dtype=np.uint8)) for n in (0, 1, 5, 300, 70000, 2)]
    stream = b"".join(ingest.frame(p) for p in payloads)
    assert stream[:4] == b"\0\0\0\0" and stream[4:8] == b"\0\0\0\x01"  # big-endian length
    for trial in range(20):
        cuts = sorted(rng.integers(0, len(stream), 12).tolist())
        rd, got, prev = ingest.LengthDelimitedReader(), [], 0
        for c in cuts + [len(stream)]:
            got += rd.feed(stream[prev:c])
            prev = c
        assert got == payloads
    rd = ingest.LengthDelimitedReader()
    assert rd.feed(stream[:3]) == [] and rd.feed(b"") == []
    with pytest.raises(ValueError):  # LengthDelimitedCodec default max_frame_length: 8 MiB
        ingest.LengthDelimitedReader().feed(struct.pack(">I", ingest.MAX_FRAME_LENGTH + 1))
    dead = ingest.LengthDelimitedReader()
    with pytest.raises(ValueError):
        dead.feed(b"\xff\xff\xff\xff")
    with pytest.raises(ValueError):
        dead.feed(ingest.frame(b"ok"))
    with pytest.raises(ValueError):
        ingest.frame(b"\0" * (ingest.MAX_FRAME_LENGTH + 1))


def test_router_rule():
    """router.rs:56-72: raw viewers get multipart items; only ids with a face-stream viewer are
    queued for inference, labelled 1280x720; a full 10-slot ring drops the frame; ConnectReq and
    garbage are ignored."""
    r = ingest.FrameRouter()
    raw, faces = [], []
    r.subscribe_raw("cam0", raw.append)
    assert r.route(ingest.encode_frame_msg("cam0", b"JPEG0")) is False  # nobody watches /face_stream: not inferred
    assert raw == [b"--frame\r\nContent-Type: image/jpeg\r\n\r\nJPEG0\r\n\r\n"] and r.infer_tx.empty()
    r.subscribe_infered("cam0", faces.append)
    assert r.route(ingest.encode_frame_msg("cam1", b"other")) is False and r.infer_tx.empty()
    assert r.route(ingest.encode_connect_req("cam0")) is False and r.route(b"\x07garbage") is False
    assert r.ignored == 2
    for i in range(12):
        queued = r.route(ingest.encode_frame_msg("cam0", b"F%d" % i))
        assert queued == (i < ingest.INFER_RING_SLOTS)
    assert r.dropped == 2 and r.infer_tx.qsize() == 10 and len(raw) == 13
    w, h, data, sender = r.infer_tx.get()
    assert (w, h, data) == (1280, 720, b"F0")
    sender(("dets", 0))
    assert faces == [("dets", 0)]
    r.unsubscribe("cam0", faces.append)  # the last viewer leaves: the stream is no longer inferred
    while not r.infer_tx.empty():
        r.infer_tx.get()
    assert r.route(ingest.encode_frame_msg("cam0", b"late")) is False and r.infer_tx.empty()


@pytest.mark.gpu
def test_wire_to_detections(oracle_lib, weights):
    """Camera bytes -> LengthDelimitedReader -> FrameRouter -> Inferer (GPU) -> the viewer's sender,
    checked against the CPU oracle; a corrupt frame and an unwatched stream on the same wire."""
    from infercam_onnx_amd import nn, synth
    from infercam_onnx_amd.inferer import Inferer
    from helpers import assert_dets_match, dets_array

    W, H = 320, 240
    pri = synth.gen_priors(W, H)
    model = nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, weights=weights, priors=pri, max_batch=4,
                              max_src=(1280, 720), det_cap=4420)
    jpegs = [synth.encode_jpeg(synth.synth_frame(77, i, 640, 480)) for i in range(5)]
    wire = ingest.frame(ingest.encode_connect_req("cam0"))
    for i, j in enumerate(jpegs):
        wire += ingest.frame(ingest.encode_frame_msg("cam0", j if i != 2 else j[:500]))
        wire += ingest.frame(ingest.encode_frame_msg("nobody_watches", j))
    router, got = ingest.FrameRouter(), []
    router.subscribe_infered("cam0", got.append)
    rd = ingest.LengthDelimitedReader()
    for off in range(0, len(wire), 4099):  # arbitrary TCP segmentation
        for payload in rd.feed(wire[off:off + 4099]):
            router.route(payload)
    assert router.infer_tx.qsize() == 5 and router.ignored == 1 and router.dropped == 0
    router.infer_tx.put(None)
    Inferer(router.infer_tx, model=model, max_batch=4).run()
    assert len(got) == 5 and got[2] == (None, nn.UFD_E_DECODE)
    for i in (0, 1, 3, 4):
        dets, st = got[i]
        assert st == 0
        ref = oracle_lib.infer_jpeg(jpegs[i], W, H, weights, pri, 0.5, 0.5)
        x = oracle_lib.normalize_nchw(oracle_lib.resize_triangle(oracle_lib.jpeg_decode_rgb(jpegs[i]), W, H))
        scores, _ = oracle_lib.forward(x, weights, pri)
        assert_dets_match(dets_array(dets), ref, scores=scores)
    model.close()
