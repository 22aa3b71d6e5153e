"""Host-side mirror of `infer_server/src/inferer.rs`: the per-frame loop that pulls JPEG slots from
the infer channel, decodes, runs the model and hands the result to the slot's sender.

The reference processes one slot at a time on one task (inferer.rs:29-50) with UltraFace-320 and
thresholds 0.5/0.5 (inferer.rs:23); here one GPU worker drains the queue in batches through the
asynchronous C ABI.  With `annotate=True` the whole iteration runs on the GPU (N1: rectangles + labels drawn
at the slot's width / height labels, JPEG quality 95 4:2:0 re-encode, inferer.rs:38-40) and the sender receives what the
reference sends -- `as_jpeg_stream_item(&buf)` (inferer.rs:41-46); otherwise it receives the detections.
"""
import collections
import queue

from .nn import UltrafaceModel, UltrafaceVariant


class Inferer:
    """`Inferer::new(infer_rx)` / `Inferer::run()`.

    `infer_rx`: a `queue.Queue` of StaticImage slots `(width, height, jpeg_bytes, sender)`
    (lib.rs:32); `sender` is a callable receiving `(detections or None, status)`; a `None` slot
    stops the loop (the reference loops forever)."""

    def __init__(self, infer_rx, model=None, max_batch=32, depth=6, annotate=False, **model_kw):
        self.infer_rx = infer_rx
        self.max_batch = max_batch
        self.annotate = annotate
        # batches in flight: the handle runs four device contexts; six to eight batches keep them fed
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
            if slots and self.annotate:
                # one annotate batch per label size (the reference's router stamps 1280 x 720 on every slot, router.rs:66-67)
                by_label = collections.OrderedDict()
                for s in slots:
                    by_label.setdefault((s[0], s[1]), []).append(s)
                for label, group in by_label.items():
                    b = self.model.prep_annotate_batch([s[2] for s in group], label, quality=95, multipart=True)
                    pending.append((self.model.submit_annotate_batch(b), group))
            elif slots:
                pending.append((self.model.submit_jpeg_batch([s[2] for s in slots]), slots))
            while len(pending) >= self.depth:
                self._deliver(*pending.popleft())
        while pending:
            self._deliver(*pending.popleft())

    def _deliver(self, ticket, slots):
        if self.annotate:
            _, status, streams = self.model.wait(ticket)
            for slot, item, st in zip(slots, streams, status):
                if slot[3] is not None and item is not None:  # (a frame that failed to decode sends nothing: inferer.rs:37)
                    slot[3](item)
            return
        results, status = self.model.wait(ticket)
        for slot, dets, st in zip(slots, results, status):
            if slot[3] is not None:
                slot[3]((dets, st))
