"""Ingest side of the hot path: the wire format the cameras speak and the router
rule that decides which frames reach the GPU worker.  Host logic only -- no sockets here; whatever
owns the TCP connection feeds bytes into `LengthDelimitedReader` and the resulting messages into
`FrameRouter.route`, whose infer queue is what `inferer.Inferer` drains.

Reference behaviour restated (nothing below is needed by the kernels):

* framing: `tokio_util::codec::LengthDelimitedCodec::new()` on both ends (`data_socket.rs:38`,
  `socket_sender.rs:68`): a 4-byte big-endian length, then that many payload bytes; frames above the
  codec's default limit of 8 MiB are an error and end the connection loop (`while let Some(Ok(..))`).
* payload: `bincode 1.3.3` of `enum ProtoMsg { ConnectReq(String), FrameMsg(FrameMsg{id: String,
  data: Vec<u8>}) }` (`common/src/protocol.rs:6-23`): u32-LE variant index, u64-LE lengths, raw
  bytes; `bincode::deserialize` tolerates trailing bytes.
* routing (`router.rs:56-72`): only `FrameMsg` is looked at (`ConnectReq` is sent by the camera,
  `socket_sender.rs:71-74`, and ignored); a raw-stream viewer of that id gets the JPEG as a multipart
  item (`lib.rs:48-57`); if the id has a face-stream viewer the frame goes to the 10-slot infer ring
  with the fixed size label 1280x720 -- `try_send_ref`, i.e. DROPPED when the ring is full.
  The reference keys its maps by `DefaultHasher(name)`; here the name itself is the key.
"""
import queue
import struct

MAX_FRAME_LENGTH = 8 * 1024 * 1024  # LengthDelimitedCodec default
INFER_RING_SLOTS = 10               # INFER_IMAGES_CHANNEL: StaticChannel<StaticImage, 10> (lib.rs:37)
STREAM_LABEL_WH = (1280, 720)       # router.rs:66-67


# ------------------------------------------------------------------ bincode ProtoMsg
def _put_bytes(b):
    return struct.pack("<Q", len(b)) + b


def encode_connect_req(name):
    """`bincode::serialize(&ProtoMsg::ConnectReq(name))`."""
    return struct.pack("<I", 0) + _put_bytes(name.encode("utf-8"))


def encode_frame_msg(stream_id, data):
    """`bincode::serialize(&ProtoMsg::FrameMsg(FrameMsg::new(id, data)))` (socket_sender.rs:85-90)."""
    return struct.pack("<I", 1) + _put_bytes(stream_id.encode("utf-8")) + _put_bytes(bytes(data))


def _take_bytes(buf, pos):
    if pos + 8 > len(buf):
        raise ValueError("unexpected end of message")
    (n,) = struct.unpack_from("<Q", buf, pos)
    pos += 8
    if n > len(buf) - pos:
        raise ValueError("length prefix beyond the message")
    return bytes(buf[pos:pos + n]), pos + n


def decode_proto_msg(buf):
    """`ProtoMsg::deserialize` (protocol.rs:26-28): -> ("connect", name) | ("frame", id, data).
    Raises ValueError where bincode returns Err (short input, unknown variant, invalid UTF-8)."""
    if len(buf) < 4:
        raise ValueError("unexpected end of message")
    (tag,) = struct.unpack_from("<I", buf, 0)
    if tag == 0:
        name, _ = _take_bytes(buf, 4)
        try:
            return ("connect", name.decode("utf-8"))
        except UnicodeDecodeError as e:
            raise ValueError("invalid utf-8 in ConnectReq") from e
    if tag == 1:
        sid, pos = _take_bytes(buf, 4)
        data, _ = _take_bytes(buf, pos)
        try:
            return ("frame", sid.decode("utf-8"), data)
        except UnicodeDecodeError as e:
            raise ValueError("invalid utf-8 in FrameMsg.id") from e
    raise ValueError("unknown ProtoMsg variant %d" % tag)


# ------------------------------------------------------------------ length-delimited framing
def frame(payload):
    """One wire frame: 4-byte big-endian length + payload."""
    if len(payload) > MAX_FRAME_LENGTH:
        raise ValueError("frame above the codec limit")
    return struct.pack(">I", len(payload)) + payload


class LengthDelimitedReader:
    """Incremental decoder of a TCP byte stream: `feed(chunk)` returns the payloads completed by
    that chunk (any split of the stream gives the same payloads).  A length above the limit raises
    ValueError and poisons the reader, as the codec error ends the reference's connection loop."""

    def __init__(self, max_frame_length=MAX_FRAME_LENGTH):
        self._buf = bytearray()
        self._max = max_frame_length
        self._dead = False

    def feed(self, chunk):
        if self._dead:
            raise ValueError("connection already failed")
        self._buf += chunk
        out = []
        while len(self._buf) >= 4:
            (n,) = struct.unpack_from(">I", self._buf, 0)
            if n > self._max:
                self._dead = True
                raise ValueError("frame of %d bytes above the codec limit" % n)
            if len(self._buf) < 4 + n:
                break
            out.append(bytes(self._buf[4:4 + n]))
            del self._buf[:4 + n]
        return out


# ------------------------------------------------------------------ multipart item + router
def as_jpeg_stream_item(data):
    """`as_jpeg_stream_item` (lib.rs:48-57): one part of the multipart/x-mixed-replace stream."""
    return b"--frame\r\nContent-Type: image/jpeg\r\n\r\n" + bytes(data) + b"\r\n\r\n"


class FrameRouter:
    """`FrameRouter::run`'s per-message rule (router.rs:56-72) over plain callables.

    `infer_tx`: a `queue.Queue(maxsize=INFER_RING_SLOTS)` of StaticImage slots
    `(width, height, jpeg_bytes, sender)` -- what `inferer.Inferer` consumes.  Viewers are callables
    registered per stream name: raw viewers receive multipart items, face-stream viewers are handed
    to the infer slot as its `sender` (the Inferer calls it with `(detections, status)`; the draw +
    re-encode step of inferer.rs:38-46 is SURVEY row N1)."""

    def __init__(self, infer_tx=None):
        self.infer_tx = infer_tx if infer_tx is not None else queue.Queue(maxsize=INFER_RING_SLOTS)
        self._raw = {}
        self._infered = {}
        self.dropped = 0   # frames lost to a full ring
        self.ignored = 0   # undecodable or non-frame messages

    def subscribe_raw(self, name, viewer):
        self._raw.setdefault(name, []).append(viewer)

    def subscribe_infered(self, name, viewer):
        self._infered.setdefault(name, []).append(viewer)

    def unsubscribe(self, name, viewer):
        for table in (self._raw, self._infered):
            if viewer in table.get(name, []):
                table[name].remove(viewer)
                if not table[name]:
                    del table[name]  # `retain(receiver_count() > 0)`: streams nobody watches are not inferred

    def route(self, payload):
        """One length-delimited payload.  Returns True if the frame was queued for inference."""
        try:
            msg = decode_proto_msg(payload)
        except ValueError:
            self.ignored += 1
            return False
        if msg[0] != "frame":
            self.ignored += 1
            return False
        _, sid, data = msg
        for viewer in self._raw.get(sid, ()):
            viewer(as_jpeg_stream_item(data))
        viewers = self._infered.get(sid)
        if not viewers:
            return False

        def sender(result, viewers=tuple(viewers)):  # broadcast::Sender: every subscriber gets the result
            for v in viewers:
                v(result)

        try:
            self.infer_tx.put_nowait((STREAM_LABEL_WH[0], STREAM_LABEL_WH[1], data, sender))
        except queue.Full:
            self.dropped += 1
            return False
        return True
