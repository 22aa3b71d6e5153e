"""Host-side mirror of the multi-stream scheduler of the C ABI (include/ufd.h, N4): the reference's
FrameRouter -> INFER_IMAGES_CHANNEL -> Inferer leg (router.rs:64-71, lib.rs:32-37, inferer.rs:29-50) for many camera
streams per GPU, each bound to a model variant and to what it wants back.  All policy lives in libufacehip.so
(csrc/sched.cpp); this module only marshals."""
import ctypes

from . import nn


def plan(queued, last, max_batch):
    """The batching rule alone (no GPU): frames taken from each stream of one batch class."""
    L = nn.load_library()
    n = len(queued)
    q = (ctypes.c_uint32 * n)(*queued)
    take = (ctypes.c_uint32 * n)()
    total = L.ufd_sched_debug_plan(q, n, last, max_batch, take)
    assert total == sum(take)
    return list(take)


class Scheduler:
    """`Scheduler(model_320=..., model_640=..., on_result=callable)`; `on_result(dict)` runs on the library's completion
    thread(s) for every frame: stream_id, tag, status, variant, replica, dets [(bbox, conf)], jpeg (bytes or None),
    batch_fill, queue_ms, total_ms.  `models_320=[...]` / `models_640=[...]` (the handles of
    `UltrafaceModel.create_replicas`, one per GPU) make it ONE scheduler over the node: streams are placed on replicas
    (`placement="round_robin"`: stream i -> replica i mod n, or "least_loaded"), each replica has its own batches in
    flight and its own completion thread."""

    def __init__(self, model_320=None, model_640=None, on_result=None, ring_slots=10, max_wait_us=2000, max_inflight=6,
                 det_cap=256, jpeg_bytes_per_frame=0, models_320=None, models_640=None, placement="round_robin"):
        self._lib = nn.load_library()
        self._models = (model_320, model_640, list(models_320 or []), list(models_640 or []))  # keep the handles alive
        self._user_cb = on_result
        self.results = []
        self._res_lock = __import__("threading").Lock()

        def _cb(_user, rp):
            r = rp.contents
            ok = r.status in (nn.UFD_OK, nn.UFD_E_TRUNCATED)
            dets = [((r.dets[i].x_tl, r.dets[i].y_tl, r.dets[i].x_br, r.dets[i].y_br), r.dets[i].conf)
                    for i in range(min(r.n, det_cap))] if ok and r.dets else None
            jpeg = ctypes.string_at(r.jpeg, r.jpeg_len) if r.jpeg and r.jpeg_len else None
            rec = dict(stream_id=r.stream_id, tag=r.tag, status=r.status, variant=r.variant, replica=r.replica, n=r.n, dets=dets,
                       jpeg=jpeg, batch_fill=r.batch_fill, queue_ms=r.queue_ms, total_ms=r.total_ms)
            if self._user_cb:
                self._user_cb(rec)
            else:
                with self._res_lock:
                    self.results.append(rec)

        self._cb = nn.UFD_RESULT_FN(_cb)  # (must outlive the scheduler)
        cfg = nn.UfdSchedConfig()
        cfg.struct_size = ctypes.sizeof(nn.UfdSchedConfig)
        cfg.model_320 = model_320._h if model_320 is not None else None
        cfg.model_640 = model_640._h if model_640 is not None else None
        self._arrays = []
        for name, lst in (("320", self._models[2]), ("640", self._models[3])):
            if lst:
                arr = (ctypes.c_void_p * len(lst))(*[m._h.value if hasattr(m._h, "value") else m._h for m in lst])
                self._arrays.append(arr)
                setattr(cfg, "models_" + name, arr)
                setattr(cfg, "n_" + name, len(lst))
        cfg.placement = {"round_robin": nn.UFD_SCHED_PLACE_ROUND_ROBIN, "least_loaded": nn.UFD_SCHED_PLACE_LEAST_LOADED}[placement]
        cfg.ring_slots, cfg.max_wait_us, cfg.max_inflight = ring_slots, max_wait_us, max_inflight
        cfg.det_cap, cfg.jpeg_bytes_per_frame = det_cap, jpeg_bytes_per_frame
        if on_result is not False:  # on_result=False: no callback at all (results are counted in the statistics only)
            cfg.on_result = self._cb
        h = ctypes.c_void_p()
        rc = self._lib.ufd_sched_create(ctypes.byref(cfg), ctypes.byref(h))
        if rc:
            raise nn.UfdError(rc, "ufd_sched_create")
        self._h = h

    def add_stream(self, stream_id, variant=320, annotate=False, label_size=(1280, 720), quality=95, multipart=False, replica=None):
        """Reference defaults: UltraFace-320 for every stream (inferer.rs:23), labels 1280 x 720 (router.rs:66-67).
        replica=None: placed by the scheduler; r: on replica r of the variant."""
        c = nn.UfdStreamConfig()
        c.struct_size = ctypes.sizeof(nn.UfdStreamConfig)
        c.stream_id, c.variant, c.annotate = int(stream_id), int(variant), int(bool(annotate))
        c.label_width, c.label_height = float(label_size[0]), float(label_size[1])
        c.quality, c.flags = int(quality), (nn.UFD_ANNOT_MULTIPART if multipart else 0)
        c.replica = 0 if replica is None else int(replica) + 1
        idx = ctypes.c_uint32()
        rc = self._lib.ufd_sched_add_stream(self._h, ctypes.byref(c), ctypes.byref(idx))
        if rc:
            raise nn.UfdError(rc, "ufd_sched_add_stream")
        return idx.value

    def remove_stream(self, stream):
        return self._lib.ufd_sched_remove_stream(self._h, stream)

    def push(self, stream, jpeg, tag=0):
        """router.rs:64-71: True if the frame found a free ring slot, False if it was dropped."""
        buf = (ctypes.c_char * len(jpeg)).from_buffer_copy(jpeg)
        rc = self._lib.ufd_sched_push(self._h, stream, buf, len(jpeg), int(tag))
        if rc == nn.UFD_E_FULL:
            return False
        if rc:
            raise nn.UfdError(rc, "ufd_sched_push")
        return True

    def push_batch(self, stream, batch, first=0):
        """`batch`: a `UltrafaceModel._prep_batch` object (pointer / length arrays of its JPEGs); pushes its frames from
        index `first` on in ONE call and returns how many were queued (the others found the ring full)."""
        acc = ctypes.c_uint32()
        n = batch.count - first
        ptrs = ctypes.byref(batch.ptrs, first * ctypes.sizeof(ctypes.c_void_p))
        lens = ctypes.byref(batch.lens, first * ctypes.sizeof(ctypes.c_size_t))
        rc = self._lib.ufd_sched_push_batch(self._h, stream, ptrs, lens, None, n, ctypes.byref(acc))
        if rc not in (nn.UFD_OK, nn.UFD_E_FULL):
            raise nn.UfdError(rc, "ufd_sched_push_batch")
        return acc.value

    def flush(self):
        self._lib.ufd_sched_flush(self._h)

    def table(self):
        """(stream-table entries in use, entries ever allocated): flat under add / remove churn."""
        live, alloc = ctypes.c_uint32(), ctypes.c_uint32()
        self._lib.ufd_sched_debug_table(self._h, ctypes.byref(live), ctypes.byref(alloc))
        return live.value, alloc.value

    def stream_replica(self, stream):
        r = ctypes.c_uint32()
        rc = self._lib.ufd_sched_stream_replica(self._h, stream, ctypes.byref(r))
        if rc:
            raise nn.UfdError(rc, "ufd_sched_stream_replica")
        return r.value

    def replica_stats(self, variant):
        """[dict(replica, streams, inflight, batches, frames)] for every replica of the variant."""
        arr = (nn.UfdSchedReplicaStats * 64)()
        n = ctypes.c_uint32()
        rc = self._lib.ufd_sched_get_replica_stats(self._h, int(variant), arr, 64, ctypes.byref(n))
        if rc:
            raise nn.UfdError(rc, "ufd_sched_get_replica_stats")
        return [dict(replica=arr[i].replica, streams=arr[i].streams, inflight=arr[i].inflight, batches=arr[i].batches,
                     frames=arr[i].frames, detections=arr[i].detections) for i in range(min(n.value, 64))]

    def stats(self):
        st = nn.UfdSchedStats()
        self._lib.ufd_sched_get_stats(self._h, ctypes.byref(st))
        return {n: getattr(st, n) for n, _ in nn.UfdSchedStats._fields_}

    def close(self):
        if self._h:
            self._lib.ufd_sched_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
