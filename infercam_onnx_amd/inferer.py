"""Host-side mirror of `infer_server/src/inferer.rs`: the per-frame loop that pulls JPEG slots from
the infer channel, decodes, runs the model and hands the result to the slot's sender.

The reference processes one slot at a time on one task (inferer.rs:29-50) with UltraFace-320 and
thresholds 0.5/0.5 (inferer.rs:23); here one GPU worker drains the queue in batches through the
asynchronous C ABI.  Draw + JPEG re-encode (inferer.rs:38-40) are outside this path:
the sender receives the detections.
"""
import collections
import queue

from .nn import UltrafaceModel, UltrafaceVariant


class Inferer:
    """`Inferer::new(infer_rx)` / `Inferer::run()`.

    `infer_rx`: a `queue.Queue` of StaticImage slots `(width, height, jpeg_bytes, sender)`
    (lib.rs:32); `sender` is a callable receiving `(detections or None, status)`; a `None` slot
    stops the loop (the reference loops forever)."""

    def __init__(self, infer_rx, model=None, max_batch=32, depth=6, **model_kw):
        self.infer_rx = infer_rx
        self.max_batch = max_batch
        # batches in flight: the handle runs three device contexts, two batches each keep them fed
        self.depth = max(1, min(depth, 8))
        # reference default: UltrafaceModel::new(W320H240, 0.5, 0.5) (inferer.rs:23)
        self.model = model or UltrafaceModel(UltrafaceVariant.W320H240, 0.5, 0.5, max_batch=max_batch, **model_kw)

    def infer_faces(self, frame):
        """`fn infer_faces(&self, frame: &RgbImage)` (inferer.rs:52-54)."""
        return self.model.run(frame)

    def _drain(self):
        slots = [self.infer_rx.get()]
        while len(slots) < self.max_batch and slots[-1] is not None:
            try:
                slots.append(self.infer_rx.get_nowait())
            except queue.Empty:
                break
        return slots

    def run(self):
        pending = collections.deque()
        stop = False
        while not stop:
            # nothing waiting to be overlapped with: hand the oldest results over before blocking
            # on the channel (a lone frame must not wait for the next one to arrive)
            if pending and self.infer_rx.empty():
                self._deliver(*pending.popleft())
                continue
            slots = self._drain()
            if slots[-1] is None:
                stop = True
                slots = slots[:-1]
            if slots:
                pending.append((self.model.submit_jpeg_batch([s[2] for s in slots]), slots))
            while len(pending) >= self.depth:
                self._deliver(*pending.popleft())
        while pending:
            self._deliver(*pending.popleft())

    def _deliver(self, ticket, slots):
        results, status = self.model.wait(ticket)
        for slot, dets, st in zip(slots, results, status):
            if slot[3] is not None:
                slot[3]((dets, st))
