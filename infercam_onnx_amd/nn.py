"""Host-side mirror of the reference's `infer_server/src/nn.rs` over the C ABI (include/ufd.h).

Same names and meaning as the reference: `Bbox`, `InferModel.run`, `UltrafaceVariant`,
`UltrafaceModel(variant, max_iou, min_confidence)`; `run(image)` returns the selected boxes with
confidences in descending order, boxes as relative `[x_tl, y_tl, x_br, y_br]` (nn.rs:12,24-26,
107-108).  All computation happens in libufacehip.so (HIP kernels for gfx950); there is no CPU
fallback: if the library or a GPU is missing, construction raises.
"""
import ctypes
import threading
import enum
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (UFD_LIBRARY: the measurement build `make EXPERIMENTS=1` leaves beside it, for tools/ab -- never a CPU stand-in)
LIB_PATH = os.environ.get("UFD_LIBRARY") or os.path.join(_HERE, "libufacehip.so")

UFD_OK = 0
UFD_E_ARG, UFD_E_DECODE, UFD_E_UNSUPPORTED, UFD_E_TRUNCATED = -1, -2, -3, -4
UFD_E_DEVICE, UFD_E_WEIGHTS, UFD_E_STATE, UFD_E_TOO_LARGE = -5, -6, -7, -8
UFD_FLAG_KEEP_LAYERS, UFD_FLAG_PROFILE, UFD_FLAG_DEVICE_ENTROPY, UFD_FLAG_HOST_ENTROPY = 1, 2, 4, 8
UFD_FLAG_TAP_LAYERS, UFD_FLAG_NO_CHAIN, UFD_FLAG_NO_RFB_SUM, UFD_FLAG_NO_STEM_FUSE = 16, 32, 64, 128
UFD_FLAG_NO_NUMA_PIN = 256
UFD_FLAG_SPIN_WAIT = 1024
UFD_FLAG_NO_RFB_TAIL = 2048
UFD_FLAG_NO_DUAL = 512
UFD_FLAG_SUBSEQ_32, UFD_FLAG_SUBSEQ_64, UFD_FLAG_TEST_DUPLICATE_DEVICES, UFD_FLAG_NO_GATE = 4096, 8192, 16384, 32768
UFD_MAX_REPLICAS = 64
UFD_SCHED_NO_WAIT = 0xFFFFFFFF
UFD_PARITY_EXACT, UFD_PARITY_LABELS_UNPINNED = 0, 1
UFD_MAX_SLOTS = 8
UFD_ANNOT_MULTIPART = 1
UFD_ANNOT_NO_TEXT = 2
UFD_E_FULL = -9

_STATUS_NAMES = {0: "UFD_OK", -1: "UFD_E_ARG", -2: "UFD_E_DECODE", -3: "UFD_E_UNSUPPORTED", -4: "UFD_E_TRUNCATED",
                 -5: "UFD_E_DEVICE", -6: "UFD_E_WEIGHTS", -7: "UFD_E_STATE", -8: "UFD_E_TOO_LARGE", -9: "UFD_E_FULL"}


class UfdDet(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("x_tl", "y_tl", "x_br", "y_br", "conf")]


class UfdConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("variant", ctypes.c_uint32), ("max_iou", ctypes.c_float),
                ("min_confidence", ctypes.c_float), ("device_id", ctypes.c_int32), ("max_batch", ctypes.c_uint32),
                ("max_src_width", ctypes.c_uint32), ("max_src_height", ctypes.c_uint32),
                ("host_threads", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("weights_path", ctypes.c_char_p),
                ("weights", ctypes.c_void_p), ("weights_floats", ctypes.c_size_t), ("priors", ctypes.c_void_p),
                ("priors_floats", ctypes.c_size_t)]


class UfdAnnotate(ctypes.Structure):
    """Arguments of the draw + re-encode step (inferer.rs:38-40): the slot's label size, quality, output buffers."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("label_width", ctypes.c_float), ("label_height", ctypes.c_float),
                ("quality", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("jpeg_out", ctypes.c_void_p),
                ("jpeg_cap", ctypes.c_size_t), ("jpeg_off", ctypes.c_void_p), ("jpeg_len", ctypes.c_void_p)]


class UfdFrameResult(ctypes.Structure):
    _fields_ = [("stream_id", ctypes.c_uint64), ("tag", ctypes.c_uint64), ("status", ctypes.c_int32), ("variant", ctypes.c_uint32),
                ("n", ctypes.c_uint32), ("batch_fill", ctypes.c_uint32), ("dets", ctypes.POINTER(UfdDet)),
                ("jpeg", ctypes.POINTER(ctypes.c_ubyte)), ("jpeg_len", ctypes.c_size_t), ("queue_ms", ctypes.c_double),
                ("total_ms", ctypes.c_double), ("replica", ctypes.c_uint32)]


UFD_RESULT_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(UfdFrameResult))


UFD_SCHED_PLACE_ROUND_ROBIN, UFD_SCHED_PLACE_LEAST_LOADED = 0, 1


class UfdSchedConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("model_320", ctypes.c_void_p), ("model_640", ctypes.c_void_p),
                ("models_320", ctypes.POINTER(ctypes.c_void_p)), ("n_320", ctypes.c_uint32),
                ("models_640", ctypes.POINTER(ctypes.c_void_p)), ("n_640", ctypes.c_uint32), ("placement", ctypes.c_uint32),
                ("ring_slots", ctypes.c_uint32), ("max_wait_us", ctypes.c_uint32), ("max_inflight", ctypes.c_uint32),
                ("det_cap", ctypes.c_uint32), ("jpeg_bytes_per_frame", ctypes.c_uint32), ("on_result", UFD_RESULT_FN),
                ("user", ctypes.c_void_p)]


class UfdStreamConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("stream_id", ctypes.c_uint64), ("variant", ctypes.c_uint32),
                ("annotate", ctypes.c_uint32), ("label_width", ctypes.c_float), ("label_height", ctypes.c_float),
                ("quality", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("replica", ctypes.c_uint32)]


class UfdSchedReplicaStats(ctypes.Structure):
    _fields_ = [("replica", ctypes.c_uint32), ("streams", ctypes.c_uint32), ("inflight", ctypes.c_uint32), ("pad", ctypes.c_uint32),
                ("batches", ctypes.c_uint64), ("frames", ctypes.c_uint64), ("detections", ctypes.c_uint64)]


class UfdSchedStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("pushed", "dropped", "delivered", "batches", "frames_in_batches", "sent_full",
                                               "sent_deadline", "sent_idle")]


class UfdKernelStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", ctypes.c_uint64), ("total_ms", ctypes.c_double),
                ("bytes", ctypes.c_double), ("flops", ctypes.c_double)]


class UfdLaunchShape(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48)] + [(n, ctypes.c_uint32) for n in (
        "workgroups", "threads", "lds_bytes", "registers", "resident_per_cu", "compute_units")]


class UfdPlanLayer(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 24)] + [(n, ctypes.c_int32) for n in (
        "kind", "leader", "ride", "chain_first", "fused_dw", "chained", "materialize", "launches", "rfb_tail", "in_tensor",
        "out_tensor", "out_coff", "tap_tensor")]


class UfdPlanTensor(ctypes.Structure):
    _fields_ = [("off_floats", ctypes.c_uint64), ("size_floats", ctypes.c_uint64)] + [(n, ctypes.c_int32) for n in (
        "c", "h", "w", "first", "last", "stored")]


class UfdHostStats(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("num_ctx", ctypes.c_uint32), ("batches", ctypes.c_uint64),
                ("launches", ctypes.c_uint64), ("wall_ms", ctypes.c_double), ("plan_ms", ctypes.c_double),
                ("copy_ms", ctypes.c_double), ("issue_ms", ctypes.c_double), ("waits", ctypes.c_uint64),
                ("wait_ms", ctypes.c_double), ("worker_busy_ms", ctypes.c_double * 8), ("gpu_batches", ctypes.c_uint64 * 8),
                ("gpu_span_ms", ctypes.c_double * 8), ("gpu_gap_ms", ctypes.c_double * 8)]


# every symbol include/ufd.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "ufd_create", "ufd_destroy", "ufd_last_error", "ufd_model_info", "ufd_infer_rgb", "ufd_infer_jpeg",
    "ufd_infer_jpeg_batch", "ufd_infer_rgb_batch", "ufd_submit_jpeg_batch", "ufd_wait", "ufd_debug_decode_jpeg",
    "ufd_debug_preproc_rgb", "ufd_debug_forward", "ufd_debug_layer_output", "ufd_debug_postproc",
    "ufd_debug_jpeg_coefficients", "ufd_debug_load_onnx", "ufd_profile_reset", "ufd_profile_sampling", "ufd_profile_read", "ufd_profile_shapes", "ufd_prime_device",
    "ufd_stage_jpeg_batch", "ufd_submit_staged", "ufd_staged_free",
    "ufd_submit_annotate_batch", "ufd_annotate_jpeg_batch", "ufd_encode_bound", "ufd_host_alloc", "ufd_host_free",
    "ufd_debug_draw_labels", "ufd_debug_encode_rgb", "ufd_model_limits",
    "ufd_sched_create", "ufd_sched_destroy", "ufd_sched_add_stream", "ufd_sched_remove_stream", "ufd_sched_push",
    "ufd_sched_flush", "ufd_sched_get_stats", "ufd_sched_debug_plan",
    "ufd_create_replicas", "ufd_model_placement", "ufd_annotate_parity", "ufd_model_host_alloc", "ufd_sched_debug_table",
    "ufd_host_stats_reset", "ufd_host_stats_read", "ufd_sched_stream_replica", "ufd_sched_get_replica_stats",
    "ufd_sched_push_batch", "ufd_debug_plan",
)

_lib = None


class UfdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (_STATUS_NAMES.get(code, "?"), code, msg))
        self.code = code


# Pinned output buffers of annotate batches (ufd_host_alloc), reused across batches: size class (power of two) -> addresses.
_pinned_free = {}
_pinned_lock = threading.Lock()
_PINNED_KEEP = 8  # buffers kept per size class


def _pinned_acquire(lib, cap):
    size = 1 << max(12, (int(cap) - 1).bit_length())
    with _pinned_lock:
        lst = _pinned_free.get(size)
        if lst:
            return lst.pop(), size
    mem = lib.ufd_host_alloc(size)
    if not mem:
        raise MemoryError("ufd_host_alloc(%d)" % size)
    return mem, size


def _pinned_release(lib, mem, size):
    with _pinned_lock:
        lst = _pinned_free.setdefault(size, [])
        if size and len(lst) < _PINNED_KEEP:
            lst.append(mem)
            return
    lib.ufd_host_free(mem)


def load_library():
    """dlopen libufacehip.so.  Raises if it was not built: the HIP path is the only path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                          "infercam_onnx_amd has no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, sz, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_int32
    pu32 = ctypes.POINTER(ctypes.c_uint32)
    L.ufd_create.argtypes = [ctypes.POINTER(UfdConfig), ctypes.POINTER(vp)]
    L.ufd_destroy.argtypes = [vp]
    L.ufd_destroy.restype = None
    L.ufd_last_error.argtypes = [vp]
    L.ufd_last_error.restype = ctypes.c_char_p
    L.ufd_model_info.argtypes = [vp, pu32, pu32, pu32]
    L.ufd_infer_rgb.argtypes = [vp, vp, u32, u32, u32, vp, u32, pu32]
    L.ufd_infer_jpeg.argtypes = [vp, vp, sz, vp, u32, pu32, pu32, pu32]
    L.ufd_infer_jpeg_batch.argtypes = [vp, vp, vp, u32, vp, u32, vp, vp]
    L.ufd_infer_rgb_batch.argtypes = [vp, vp, u32, u32, u32, u32, vp, u32, vp]
    L.ufd_submit_jpeg_batch.argtypes = [vp, vp, vp, u32, vp, u32, vp, vp, pu32]
    L.ufd_wait.argtypes = [vp, u32]
    L.ufd_stage_jpeg_batch.argtypes = [vp, vp, vp, u32, ctypes.POINTER(vp)]
    L.ufd_submit_staged.argtypes = [vp, vp, vp, u32, vp, vp, pu32]
    L.ufd_staged_free.argtypes = [vp, vp]
    L.ufd_staged_free.restype = None
    L.ufd_debug_decode_jpeg.argtypes = [vp, vp, sz, vp, sz, pu32, pu32]
    L.ufd_debug_preproc_rgb.argtypes = [vp, vp, u32, u32, u32, vp]
    L.ufd_debug_forward.argtypes = [vp, vp, u32, vp, vp]
    L.ufd_debug_layer_output.argtypes = [vp, u32, u32, vp, sz, ctypes.POINTER(sz)]
    L.ufd_debug_postproc.argtypes = [vp, vp, vp, u32, vp, u32, vp]
    L.ufd_debug_jpeg_coefficients.argtypes = [vp, sz, vp, sz, pu32, pu32, pu32]
    L.ufd_debug_load_onnx.argtypes = [ctypes.c_char_p, u32, vp, sz, vp, sz, pu32, ctypes.c_char_p, sz]
    L.ufd_submit_annotate_batch.argtypes = [vp, vp, vp, u32, ctypes.POINTER(UfdAnnotate), vp, u32, vp, vp, pu32]
    L.ufd_annotate_jpeg_batch.argtypes = [vp, vp, vp, u32, ctypes.POINTER(UfdAnnotate), vp, u32, vp, vp]
    L.ufd_encode_bound.argtypes = [u32, u32]
    L.ufd_encode_bound.restype = sz
    L.ufd_host_alloc.argtypes = [sz]
    L.ufd_host_alloc.restype = vp
    L.ufd_host_free.argtypes = [vp]
    L.ufd_host_free.restype = None
    L.ufd_debug_draw_labels.argtypes = [vp, vp, u32, u32, u32, vp, u32, ctypes.c_float, ctypes.c_float, u32]
    L.ufd_debug_encode_rgb.argtypes = [vp, vp, u32, u32, u32, u32, u32, vp, sz, ctypes.POINTER(sz)]
    L.ufd_model_limits.argtypes = [vp, pu32, pu32, pu32]
    L.ufd_create_replicas.argtypes = [ctypes.POINTER(UfdConfig), ctypes.POINTER(i32), u32, ctypes.POINTER(vp)]
    L.ufd_model_placement.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), pu32, ctypes.c_char_p, sz, ctypes.c_char_p, sz]
    L.ufd_annotate_parity.argtypes = [u32]
    L.ufd_model_host_alloc.argtypes = [vp, sz]
    L.ufd_model_host_alloc.restype = vp
    L.ufd_sched_debug_table.argtypes = [vp, pu32, pu32]
    L.ufd_sched_create.argtypes = [ctypes.POINTER(UfdSchedConfig), ctypes.POINTER(vp)]
    L.ufd_sched_destroy.argtypes = [vp]
    L.ufd_sched_destroy.restype = None
    L.ufd_sched_add_stream.argtypes = [vp, ctypes.POINTER(UfdStreamConfig), pu32]
    L.ufd_sched_remove_stream.argtypes = [vp, u32]
    L.ufd_sched_push.argtypes = [vp, u32, vp, sz, ctypes.c_uint64]
    L.ufd_sched_flush.argtypes = [vp]
    L.ufd_sched_get_stats.argtypes = [vp, ctypes.POINTER(UfdSchedStats)]
    L.ufd_sched_push_batch.argtypes = [vp, u32, vp, vp, vp, u32, pu32]
    L.ufd_sched_stream_replica.argtypes = [vp, u32, pu32]
    L.ufd_sched_get_replica_stats.argtypes = [vp, u32, ctypes.POINTER(UfdSchedReplicaStats), u32, pu32]
    L.ufd_sched_debug_plan.argtypes = [pu32, u32, u32, u32, pu32]
    L.ufd_sched_debug_plan.restype = u32
    L.ufd_profile_reset.argtypes = [vp]
    L.ufd_profile_sampling.argtypes = [vp, u32]
    L.ufd_profile_read.argtypes = [vp, vp, u32, pu32]
    L.ufd_profile_shapes.argtypes = [vp, vp, u32, pu32]
    L.ufd_prime_device.argtypes = [i32]
    L.ufd_debug_plan.argtypes = [u32, u32, u32, ctypes.POINTER(UfdPlanLayer), u32, pu32, ctypes.POINTER(UfdPlanTensor), u32, pu32,
                                 ctypes.POINTER(ctypes.c_uint64)]
    L.ufd_host_stats_reset.argtypes = [vp]
    L.ufd_host_stats_read.argtypes = [vp, ctypes.POINTER(UfdHostStats)]
    _lib = L
    return L


def debug_plan(variant, max_batch, flags=0):
    """The launch plan of the network alone (ufd_debug_plan; no GPU needed): -> (layers, tensors, arena_floats) as dicts."""
    L = load_library()
    layers, tensors = (UfdPlanLayer * 64)(), (UfdPlanTensor * 128)()
    nl, nt, arena = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
    rc = L.ufd_debug_plan(int(variant), int(max_batch), int(flags), layers, 64, ctypes.byref(nl), tensors, 128, ctypes.byref(nt),
                          ctypes.byref(arena))
    if rc:
        raise UfdError(rc, "ufd_debug_plan")
    ls = [dict(name=layers[i].name.decode(), **{f: getattr(layers[i], f) for f, _ in UfdPlanLayer._fields_[1:]}) for i in range(nl.value)]
    ts = [{f: getattr(tensors[i], f) for f, _ in UfdPlanTensor._fields_} for i in range(nt.value)]
    return ls, ts, arena.value


def prime_device(device_id=0):
    """ufd_prime_device: opens this library's hardware queues on the device BEFORE another copy of the HIP runtime in the
    process (torch's) touches it -- a handle created behind torch's first copy or kernel runs 21 % slower for good
    (include/ufd.h).  In a Python host that also uses torch on the GPU: `torch.cuda.set_device(d)` (initialises torch's
    runtime without opening a queue), then this, then everything else -- with the two the other way round torch's runtime no
    longer finds the device."""
    rc = load_library().ufd_prime_device(int(device_id))
    if rc:
        raise UfdError(rc, (load_library().ufd_last_error(None) or b"").decode())


def jpeg_coefficients(jpeg):
    """Host Huffman stage alone (no GPU): (int16 coefficients, width, height)."""
    L = load_library()
    buf = (ctypes.c_char * len(jpeg)).from_buffer_copy(jpeg)
    n, w, h = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    rc = L.ufd_debug_jpeg_coefficients(buf, len(jpeg), None, 0, ctypes.byref(n), ctypes.byref(w), ctypes.byref(h))
    if rc:
        raise UfdError(rc, "jpeg header")
    coef = np.empty(n.value, np.int16)
    rc = L.ufd_debug_jpeg_coefficients(buf, len(jpeg), coef.ctypes.data, coef.size, ctypes.byref(n), ctypes.byref(w),
                                       ctypes.byref(h))
    if rc:
        raise UfdError(rc, "jpeg entropy decode")
    return coef, w.value, h.value


def load_onnx(path, variant=640):
    """get_model's parsing step alone (host only): .onnx -> (packed weight blob, priors or None)."""
    from . import topology as T

    L = load_library()
    w, h = T.VARIANTS[variant]
    blob = np.empty(T.total_weight_floats(), np.float32)
    pri = np.empty((T.num_priors(w, h), 4), np.float32)
    found = ctypes.c_uint32()
    err = ctypes.create_string_buffer(256)
    rc = L.ufd_debug_load_onnx(os.fsencode(path), variant, blob.ctypes.data, blob.size, pri.ctypes.data, pri.size,
                               ctypes.byref(found), err, 256)
    if rc:
        raise UfdError(rc, err.value.decode())
    return blob, (pri if found.value else None)


#: Bounding box `[x_top_left, y_top_left, x_bottom_right, y_bottom_right]` (nn.rs:12)
Bbox = tuple


class InferModel:
    """`pub trait InferModel { fn run(&self, input: &RgbImage) -> Result<Vec<(Bbox, f32)>>; }` (nn.rs:24-26)"""

    def run(self, image):
        raise NotImplementedError


class UltrafaceVariant(enum.Enum):
    """Supported variants of the Ultraface model (nn.rs:29-42)."""
    W640H480 = 640
    W320H240 = 320

    def width_height(self):
        return (640, 480) if self is UltrafaceVariant.W640H480 else (320, 240)


def _dets_to_list(arr, n, first=0):
    """[(bbox, conf)] of detections first .. first + n of a UfdDet array (five f32 each: one numpy view, no ctypes object
    per detection or per field)."""
    if n <= 0:
        return []
    a = np.frombuffer(arr, dtype=np.float32, count=int(n) * 5, offset=int(first) * 20).reshape(int(n), 5).tolist()
    return [((r[0], r[1], r[2], r[3]), r[4]) for r in a]


class UltrafaceModel(InferModel):
    """Loaded Ultraface model resident on one MI355X, with post-processing thresholds (nn.rs:45-67).

    `UltrafaceModel(variant, max_iou, min_confidence)` as in the reference; keyword arguments add
    placement (`device_id`, `max_batch`) and the weight source (`weights_path` = .onnx, or a packed
    f32 `weights` blob + optional `priors`; default = the reference's cache path, nn.rs:144-156).
    Entropy stage: by default the GPU kernels decode baseline single-scan streams (with or without
    restart markers) and the host workers progressive / multi-scan files; `host_entropy=True` keeps
    the whole Huffman stage on the host workers (`device_entropy` is accepted and ignored).
    """

    def __init__(self, variant, max_iou, min_confidence, *, device_id=0, max_batch=1, weights=None, priors=None,
                 weights_path=None, max_src=(0, 0), host_threads=0, keep_layers=False, profile=False, det_cap=1024,
                 device_entropy=False, host_entropy=False, tap_layers=False, no_chain=False, no_rfb_sum=False,
                 no_stem_fuse=False, no_numa_pin=False, no_dual=False, extra_flags=0, _handle=None):
        self._h = None
        self._lib = load_library()
        self.variant = variant
        self.width, self.height = variant.width_height()
        self.max_iou, self.min_confidence = float(max_iou), float(min_confidence)
        self.max_batch = int(max_batch)
        self.det_cap = int(det_cap)
        if _handle is not None:  # a handle ufd_create_replicas made
            h = _handle
        else:
            cfg, _keep = self._config(variant, max_iou, min_confidence, device_id=device_id, max_batch=max_batch, weights=weights,
                                      priors=priors, weights_path=weights_path, max_src=max_src, host_threads=host_threads,
                                      keep_layers=keep_layers, profile=profile, device_entropy=device_entropy,
                                      host_entropy=host_entropy, tap_layers=tap_layers, no_chain=no_chain, no_rfb_sum=no_rfb_sum,
                                      no_stem_fuse=no_stem_fuse, no_numa_pin=no_numa_pin, no_dual=no_dual, extra_flags=extra_flags)
            h = ctypes.c_void_p()
            rc = self._lib.ufd_create(ctypes.byref(cfg), ctypes.byref(h))
            if rc != UFD_OK:
                raise UfdError(rc, (self._lib.ufd_last_error(None) or b"").decode())
        self._h = h
        k = ctypes.c_uint32()
        self._lib.ufd_model_info(h, None, None, ctypes.byref(k))
        self.num_priors = k.value
        self._pending = {}

    @staticmethod
    def _config(variant, max_iou, min_confidence, *, device_id=0, max_batch=1, weights=None, priors=None, weights_path=None,
                max_src=(0, 0), host_threads=0, keep_layers=False, profile=False, device_entropy=False, host_entropy=False,
                tap_layers=False, no_chain=False, no_rfb_sum=False, no_stem_fuse=False, no_numa_pin=False, no_dual=False,
                extra_flags=0):
        """ufd_config of UltrafaceModel::new's arguments; returns (cfg, arrays the cfg points into)."""
        cfg = UfdConfig()
        cfg.struct_size = ctypes.sizeof(UfdConfig)
        cfg.variant = variant.value
        cfg.max_iou, cfg.min_confidence = float(max_iou), float(min_confidence)
        cfg.device_id, cfg.max_batch = int(device_id), int(max_batch)
        cfg.max_src_width, cfg.max_src_height = int(max_src[0]), int(max_src[1])
        cfg.host_threads = int(host_threads)
        cfg.flags = ((UFD_FLAG_KEEP_LAYERS if keep_layers else 0) | (UFD_FLAG_PROFILE if profile else 0) |
                     (UFD_FLAG_DEVICE_ENTROPY if device_entropy else 0) | (UFD_FLAG_HOST_ENTROPY if host_entropy else 0) |
                     (UFD_FLAG_TAP_LAYERS if tap_layers else 0) | (UFD_FLAG_NO_CHAIN if no_chain else 0) |
                     (UFD_FLAG_NO_RFB_SUM if no_rfb_sum else 0) | (UFD_FLAG_NO_STEM_FUSE if no_stem_fuse else 0) |
                     (UFD_FLAG_NO_NUMA_PIN if no_numa_pin else 0) | (UFD_FLAG_NO_DUAL if no_dual else 0) | int(extra_flags))
        keep = []
        if weights is not None:
            w = np.ascontiguousarray(weights, np.float32).ravel()
            keep.append(w)
            cfg.weights, cfg.weights_floats = w.ctypes.data, w.size
            if priors is not None:
                p = np.ascontiguousarray(priors, np.float32).ravel()
                keep.append(p)
                cfg.priors, cfg.priors_floats = p.ctypes.data, p.size
        elif weights_path is not None:
            keep.append(os.fsencode(weights_path))
            cfg.weights_path = keep[-1]
        return cfg, keep

    @classmethod
    def create_replicas(cls, variant, max_iou, min_confidence, device_ids, *, det_cap=1024, **kw):
        """One handle per GPU of `device_ids` from ONE process (ufd_create_replicas): the weight source is read once,
        device_ids[0]'s packed image is broadcast to the others over RCCL.  Returns [UltrafaceModel] in device order."""
        lib = load_library()
        cfg, _keep = cls._config(variant, max_iou, min_confidence, **kw)
        ids = (ctypes.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
        hs = (ctypes.c_void_p * max(len(device_ids), 1))()
        rc = lib.ufd_create_replicas(ctypes.byref(cfg), ids, len(device_ids), hs)
        if rc != UFD_OK:
            raise UfdError(rc, (lib.ufd_last_error(None) or b"").decode())
        return [cls(variant, max_iou, min_confidence, max_batch=kw.get("max_batch", 1), det_cap=det_cap,
                    _handle=ctypes.c_void_p(hs[i])) for i in range(len(device_ids))]

    def placement(self):
        """Where the handle lives: {device_id, pci, numa_node, pinned_cpus, cpu_list} (ufd_model_placement)."""
        dev, node, n = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_uint32()
        pci, cpus = ctypes.create_string_buffer(64), ctypes.create_string_buffer(1024)
        self._check(self._lib.ufd_model_placement(self._h, ctypes.byref(dev), ctypes.byref(node), ctypes.byref(n), pci, 64, cpus, 1024))
        return {"device_id": dev.value, "pci": pci.value.decode(), "numa_node": node.value, "pinned_cpus": n.value,
                "cpu_list": cpus.value.decode()}

    # -- lifetime
    def close(self):
        if self._h:
            self._lib.ufd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, allow=()):
        if rc != UFD_OK and rc not in allow:
            raise UfdError(rc, (self._lib.ufd_last_error(self._h) or b"").decode())
        return rc

    # -- InferModel::run (nn.rs:178-186)
    def run(self, image):
        """image: HxWx3 uint8 RGB (any size) -> [(bbox, confidence)] in descending confidence."""
        return self.run_batch(np.asarray(image)[None])[0]

    def run_batch(self, images):
        """images: [N,H,W,3] uint8, all the same size."""
        imgs = np.ascontiguousarray(images, np.uint8)
        n, h, w, c = imgs.shape
        assert c == 3
        out = (UfdDet * (n * self.det_cap))()
        cnt = (ctypes.c_uint32 * n)()
        self._check(self._lib.ufd_infer_rgb_batch(self._h, imgs.ctypes.data, w, h, w * 3, n, out, self.det_cap, cnt),
                    allow=(UFD_E_TRUNCATED,))
        return [_dets_to_list(out, min(cnt[i], self.det_cap), i * self.det_cap) for i in range(n)]

    # -- Inferer::run decode -> infer (inferer.rs:35-37)
    def infer_jpeg(self, jpeg):
        out = (UfdDet * self.det_cap)()
        n, w, h = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        buf = jpeg if isinstance(jpeg, bytes) else (ctypes.c_char * len(jpeg)).from_buffer_copy(jpeg)  # (bytes: passed as they lie)
        self._check(self._lib.ufd_infer_jpeg(self._h, buf, len(jpeg), out, self.det_cap, ctypes.byref(n),
                                             ctypes.byref(w), ctypes.byref(h)), allow=(UFD_E_TRUNCATED,))
        return _dets_to_list(out, min(n.value, self.det_cap))

    class _Batch:
        __slots__ = ("bufs", "ptrs", "lens", "out", "cnt", "status", "count", "ticket", "staged")

    def _prep_batch(self, jpegs):
        b = UltrafaceModel._Batch()
        b.count = len(jpegs)
        b.bufs = [j if isinstance(j, (ctypes.Array,)) else (ctypes.c_char * len(j)).from_buffer_copy(j) for j in jpegs]
        b.ptrs = (ctypes.c_void_p * b.count)(*[ctypes.addressof(x) for x in b.bufs])
        b.lens = (ctypes.c_size_t * b.count)(*[len(x) for x in b.bufs])
        b.out = (UfdDet * (b.count * self.det_cap))()
        b.cnt = (ctypes.c_uint32 * b.count)()
        b.status = (ctypes.c_int32 * b.count)()
        return b

    def _collect(self, b):
        res = []
        for i in range(b.count):
            if b.status[i] in (UFD_OK, UFD_E_TRUNCATED):
                res.append(_dets_to_list(b.out, min(b.cnt[i], self.det_cap), i * self.det_cap))
            else:
                res.append(None)  # frame skipped (corrupt / unsupported)
        return res, list(b.status)

    def infer_jpeg_batch(self, jpegs):
        """-> ([detections or None per frame], [status per frame])"""
        b = self._prep_batch(jpegs)
        self._check(self._lib.ufd_infer_jpeg_batch(self._h, b.ptrs, b.lens, b.count, b.out, self.det_cap, b.cnt,
                                                   b.status))
        return self._collect(b)

    def submit_jpeg_batch(self, jpegs):
        """Asynchronous form: returns a ticket for `wait`."""
        b = jpegs if isinstance(jpegs, UltrafaceModel._Batch) else self._prep_batch(jpegs)
        t = ctypes.c_uint32()
        self._check(self._lib.ufd_submit_jpeg_batch(self._h, b.ptrs, b.lens, b.count, b.out, self.det_cap, b.cnt,
                                                    b.status, ctypes.byref(t)))
        b.ticket = t.value
        self._pending[t.value] = b
        return t.value

    def stage_jpeg_batch(self, jpegs):
        """Places a batch in HBM (headers parsed, bytes uploaded): -> staged batch for `submit_staged`.
        Not available with host_entropy=True.  Free with `free_staged` after the last wait."""
        b = self._prep_batch(jpegs)
        h = ctypes.c_void_p()
        self._check(self._lib.ufd_stage_jpeg_batch(self._h, b.ptrs, b.lens, b.count, ctypes.byref(h)))
        b.staged = h
        return b

    def submit_staged(self, b):
        """Runs the path on a staged (HBM-resident) batch; returns a ticket for `wait`.  A staged
        batch can be in flight once at a time (its output arrays are reused)."""
        t = ctypes.c_uint32()
        self._check(self._lib.ufd_submit_staged(self._h, b.staged, b.out, self.det_cap, b.cnt, b.status, ctypes.byref(t)))
        b.ticket = t.value
        self._pending[t.value] = b
        return t.value

    def free_staged(self, b):
        if getattr(b, "staged", None):
            self._lib.ufd_staged_free(self._h, b.staged)
            b.staged = None

    def wait(self, ticket, collect=True):
        b = self._pending.pop(ticket)
        self._check(self._lib.ufd_wait(self._h, ticket))
        if getattr(b, "annot", None) is not None:
            return self._collect_annot(b) if collect else (b.cnt, b.status, b.jpeg_len)
        return self._collect(b) if collect else (b.cnt, b.status)

    # -- N1: the rest of the Inferer::run iteration (inferer.rs:38-46): rectangles + JPEG re-encode on the GPU
    class _AnnotBatch(_Batch):
        __slots__ = ("annot", "jpeg_buf", "jpeg_mem", "jpeg_off", "jpeg_len", "owner", "jpeg_alloc")

        def __del__(self):
            mem, owner = getattr(self, "jpeg_mem", None), getattr(self, "owner", None)
            if mem and owner is not None:
                _pinned_release(owner, mem, getattr(self, "jpeg_alloc", 0))
                self.jpeg_mem = None

    def prep_annotate_batch(self, jpegs, label_size, quality=95, multipart=False, out_bytes_per_frame=None, text=True, pinned=True,
                            cap_bytes=None):
        """Buffers of one annotate batch (reusable): pinned output memory from ufd_host_alloc (the batch's own chain writes
        the finished streams there); pinned=False: ordinary memory, fetched by a copy in ufd_wait.  label_size = the
        slot's (width, height) -- 1280 x 720 in the reference's router, whatever the JPEG's size (router.rs:66-67)."""
        base = self._prep_batch(jpegs)
        b = UltrafaceModel._AnnotBatch()
        for k in UltrafaceModel._Batch.__slots__:
            if hasattr(base, k):
                setattr(b, k, getattr(base, k))
        if out_bytes_per_frame is None and cap_bytes is None:  # worst case of the largest frame in the batch
            out_bytes_per_frame = 0
            for x in b.bufs:
                try:
                    _, w, h = jpeg_coefficients_header(bytes(x))
                except UfdError:
                    continue
                out_bytes_per_frame = max(out_bytes_per_frame, self._lib.ufd_encode_bound(w, h))
        cap = max(int(out_bytes_per_frame), 1024) * b.count if cap_bytes is None else int(cap_bytes)  # (cap_bytes: the exact size, tests)
        b.owner = self._lib
        if pinned:
            # (from a small pool: pinning memory costs a millisecond, and the reference's ring slots are reused too, lib.rs:32-37)
            b.jpeg_mem, b.jpeg_alloc = _pinned_acquire(self._lib, cap)
            b.jpeg_buf = (ctypes.c_ubyte * cap).from_address(b.jpeg_mem)
        else:
            b.jpeg_mem = None
            b.jpeg_buf = (ctypes.c_ubyte * cap)()
        b.jpeg_off = (ctypes.c_size_t * b.count)()
        b.jpeg_len = (ctypes.c_size_t * b.count)()
        a = UfdAnnotate()
        a.struct_size = ctypes.sizeof(UfdAnnotate)
        a.label_width, a.label_height = float(label_size[0]), float(label_size[1])
        a.quality, a.flags = int(quality), (UFD_ANNOT_MULTIPART if multipart else 0) | (0 if text else UFD_ANNOT_NO_TEXT)
        a.jpeg_out, a.jpeg_cap = (b.jpeg_mem if pinned else ctypes.addressof(b.jpeg_buf)), cap
        a.jpeg_off, a.jpeg_len = ctypes.addressof(b.jpeg_off), ctypes.addressof(b.jpeg_len)
        b.annot = a
        return b

    def submit_annotate_batch(self, b):
        t = ctypes.c_uint32()
        self._check(self._lib.ufd_submit_annotate_batch(self._h, b.ptrs, b.lens, b.count, ctypes.byref(b.annot), b.out,
                                                        self.det_cap, b.cnt, b.status, ctypes.byref(t)))
        b.ticket = t.value
        self._pending[t.value] = b
        return t.value

    def _collect_annot(self, b):
        dets, status = self._collect(b)
        view = memoryview(b.jpeg_buf)
        streams = [bytes(view[b.jpeg_off[i]:b.jpeg_off[i] + b.jpeg_len[i]]) if b.jpeg_len[i] else None for i in range(b.count)]
        return dets, status, streams

    def annotate_jpeg_batch(self, jpegs, label_size, quality=95, multipart=False, text=True):
        """decode -> infer -> rectangles + labels -> re-encode for a batch: ([detections], [status], [annotated JPEG bytes or None])."""
        b = self.prep_annotate_batch(jpegs, label_size, quality, multipart, text=text)
        return self.wait(self.submit_annotate_batch(b))

    def annotate_jpeg(self, jpeg, label_size, quality=95, multipart=False, text=True):
        dets, status, streams = self.annotate_jpeg_batch([jpeg], label_size, quality, multipart, text=text)
        if status[0] not in (UFD_OK, UFD_E_TRUNCATED):
            raise UfdError(status[0], "frame skipped")
        return dets[0], streams[0]

    def debug_draw_labels(self, rgb, dets, label_size, text=True):
        """N1 drawing stage alone on the GPU: HxWx3 uint8 + [(bbox, conf)] or [n,5] -> copy with rectangles (+ labels)."""
        out = np.ascontiguousarray(rgb, np.uint8).copy()
        h, w, _ = out.shape
        if not isinstance(dets, np.ndarray):
            dets = [list(b) + [c] for b, c in dets]
        d = np.ascontiguousarray(np.asarray(dets, np.float32).reshape(-1, 5))
        self._check(self._lib.ufd_debug_draw_labels(self._h, out.ctypes.data, w, h, w * 3, d.ctypes.data, len(d),
                                                    float(label_size[0]), float(label_size[1]), int(bool(text))))
        return out

    def debug_encode_rgb(self, rgb, quality=95, multipart=False):
        """N1 encoder alone on the GPU: HxWx3 uint8 -> JPEG bytes (turbojpeg::compress_image(.., quality, Sub2x2))."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w, _ = rgb.shape
        cap = self._lib.ufd_encode_bound(w, h)
        out = np.empty(cap, np.uint8)
        n = ctypes.c_size_t()
        self._check(self._lib.ufd_debug_encode_rgb(self._h, rgb.ctypes.data, w, h, w * 3, int(quality),
                                                   UFD_ANNOT_MULTIPART if multipart else 0, out.ctypes.data, cap, ctypes.byref(n)))
        return out[:n.value].tobytes()

    # -- stage taps used by the parity tests
    def debug_decode_jpeg(self, jpeg):
        """A1 on the GPU: JPEG -> HxWx3 uint8."""
        buf = (ctypes.c_char * len(jpeg)).from_buffer_copy(jpeg)
        _, w, h = jpeg_coefficients_header(jpeg)
        out = np.empty((h, w, 3), np.uint8)
        ww, hh = ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self._lib.ufd_debug_decode_jpeg(self._h, buf, len(jpeg), out.ctypes.data, out.nbytes,
                                                    ctypes.byref(ww), ctypes.byref(hh)))
        return out

    def debug_preproc(self, rgb):
        """A2-A4 on the GPU: HxWx3 uint8 -> [3, H_model, W_model] f32."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w, _ = rgb.shape
        out = np.empty((3, self.height, self.width), np.float32)
        self._check(self._lib.ufd_debug_preproc_rgb(self._h, rgb.ctypes.data, w, h, w * 3, out.ctypes.data))
        return out

    def debug_forward(self, inputs):
        """A6 on the GPU: [N,3,H,W] f32 -> (scores [N,K,2], boxes [N,K,4])."""
        x = np.ascontiguousarray(inputs, np.float32)
        n = x.shape[0]
        scores = np.empty((n, self.num_priors, 2), np.float32)
        boxes = np.empty((n, self.num_priors, 4), np.float32)
        self._check(self._lib.ufd_debug_forward(self._h, x.ctypes.data, n, scores.ctypes.data, boxes.ctypes.data))
        return scores, boxes

    def debug_layer_output(self, layer, frame=0):
        """Output of conv `layer` for `frame` of the last forward (keep_layers: the unfused plan;
        tap_layers: the issued plan -- raises UFD_E_STATE for a layer fused into the next launch)."""
        nf = ctypes.c_size_t()
        self._check(self._lib.ufd_debug_layer_output(self._h, layer, frame, None, 0, ctypes.byref(nf)), allow=(UFD_E_ARG,))
        out = np.empty(nf.value, np.float32)
        self._check(self._lib.ufd_debug_layer_output(self._h, layer, frame, out.ctypes.data, out.size, ctypes.byref(nf)))
        return out

    def debug_postproc(self, scores, boxes):
        """A7-A10 on the GPU: raw outputs [N,K,2], [N,K,4] -> per-frame detection lists."""
        s = np.ascontiguousarray(scores, np.float32)
        b = np.ascontiguousarray(boxes, np.float32)
        n = s.shape[0]
        cap = self.num_priors
        out = (UfdDet * (n * cap))()
        cnt = (ctypes.c_uint32 * n)()
        self._check(self._lib.ufd_debug_postproc(self._h, s.ctypes.data, b.ctypes.data, n, out, cap, cnt))
        return [_dets_to_list(out, cnt[i], i * cap) for i in range(n)]

    # -- measurement
    def profile_reset(self):
        self._check(self._lib.ufd_profile_reset(self._h))

    def profile_sampling(self, every_n):
        """Record kernel events only for every `every_n`-th batch."""
        self._check(self._lib.ufd_profile_sampling(self._h, int(every_n)))

    def host_stats_reset(self):
        self._check(self._lib.ufd_host_stats_reset(self._h))

    def host_stats(self):
        """What the host side of the asynchronous pipeline cost since the last reset (ufd_host_stats, include/ufd.h):
        per-batch microseconds of header scan / staging copy / launch issue / caller wait, the busy share of every
        issue worker, and per context the device-time span of its batches and the idle gaps between them."""
        st = UfdHostStats()
        st.struct_size = ctypes.sizeof(UfdHostStats)
        self._check(self._lib.ufd_host_stats_read(self._h, ctypes.byref(st)))
        nb = max(int(st.batches), 1)
        nc = int(st.num_ctx)
        wall = max(st.wall_ms, 1e-9)
        gaps = [max(int(st.gpu_batches[c]) - 1, 1) for c in range(nc)]
        return {
            "batches": int(st.batches), "wall_ms": round(st.wall_ms, 3),
            "launches_per_batch": round(st.launches / nb, 1),
            "per_batch_us": {"header_scan": round(st.plan_ms / nb * 1e3, 1), "staging_memcpy": round(st.copy_ms / nb * 1e3, 1),
                             "launch_issue": round(st.issue_ms / nb * 1e3, 1),
                             "wait": round(st.wait_ms / max(int(st.waits), 1) * 1e3, 1)},
            "worker_busy_share": [round(st.worker_busy_ms[c] / wall, 4) for c in range(nc)],
            "gpu_span_share": [round(st.gpu_span_ms[c] / wall, 4) for c in range(nc)],
            "gpu_span_ms_per_batch": [round(st.gpu_span_ms[c] / max(int(st.gpu_batches[c]), 1), 4) for c in range(nc)],
            "gpu_idle_gap_us_per_batch": [round(st.gpu_gap_ms[c] / gaps[c] * 1e3, 1) for c in range(nc)],
        }

    def profile_read(self):
        """-> [dict(name, launches, total_ms, bytes, flops)] per kernel since the last reset."""
        cap = 256
        arr = (UfdKernelStat * cap)()
        n = ctypes.c_uint32()
        self._check(self._lib.ufd_profile_read(self._h, arr, cap, ctypes.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), total_ms=arr[i].total_ms,
                     bytes=arr[i].bytes, flops=arr[i].flops) for i in range(min(n.value, cap))]


    def profile_shapes(self):
        """-> [dict] per profiled label: workgroups, threads, lds_bytes, registers of its last launch, workgroups a CU holds
        at once (the runtime's occupancy query), and from them slots / rounds / last-round fill on this GPU's CUs."""
        cap = 256
        arr = (UfdLaunchShape * cap)()
        n = ctypes.c_uint32()
        self._check(self._lib.ufd_profile_shapes(self._h, arr, cap, ctypes.byref(n)))
        out = []
        for i in range(min(n.value, cap)):
            a = arr[i]
            slots = a.compute_units * max(a.resident_per_cu, 1)
            full, rem = divmod(a.workgroups, slots)
            out.append(dict(name=a.name.decode(), workgroups=int(a.workgroups), threads=int(a.threads), lds_bytes=int(a.lds_bytes),
                            registers=int(a.registers), resident_per_cu=int(a.resident_per_cu), slots=int(slots),
                            rounds=round(a.workgroups / slots, 3),
                            last_round_fill=round((rem if rem else (slots if full else 0)) / slots, 3)))
        return out


def jpeg_coefficients_header(jpeg):
    """(n_int16, width, height) from the JPEG header (host only)."""
    L = load_library()
    buf = (ctypes.c_char * len(jpeg)).from_buffer_copy(jpeg)
    n, w, h = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    rc = L.ufd_debug_jpeg_coefficients(buf, len(jpeg), None, 0, ctypes.byref(n), ctypes.byref(w), ctypes.byref(h))
    if rc:
        raise UfdError(rc, "jpeg header")
    return n.value, w.value, h.value
