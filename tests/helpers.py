"""Shared helpers for the parity tests."""
import numpy as np


def dets_array(dets):
    """[(bbox, conf)] -> [n,5] f32"""
    return np.array([list(b) + [c] for b, c in dets], np.float32).reshape(-1, 5)


def _iou(a, b):
    """nn.rs:227-243 in float64 (only used to recognise borderline NMS decisions)"""
    def area(x):
        w, h = x[3] - x[1], x[2] - x[0]
        return 0.0 if (w < 0 or h < 0) else w * h
    o = [max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])]
    ov = area(o)
    return ov / (area(a) + area(b) - ov + 1e-7)


def assert_dets_match(got, ref, scores=None, min_conf=0.5, max_iou=0.5, atol=1e-4, what=""):
    """Detection lists must agree within `atol` (north_star bar: 1e-3).

    1. Same length and element-wise close (order included): pass.
    2. Otherwise the lists are matched as sets (fp32 rounding can swap two detections whose
       confidences differ by < atol).  A detection without a partner is excused only when the
       decision that produced it is provably borderline at fp32 resolution:
         - its confidence is within atol of the threshold (strict `>` flipped), or
         - its IoU with some detection of the other list is within 1e-3 of max_iou (NMS flipped),
       and such leftovers must stay below 1 % of the list."""
    got, ref = np.asarray(got, np.float32).reshape(-1, 5), np.asarray(ref, np.float32).reshape(-1, 5)
    if got.shape == ref.shape and (got.size == 0 or np.abs(got - ref).max() <= atol):
        return 0
    used = np.zeros(len(ref), bool)
    left_got = []
    for g in got:
        if len(ref):
            d = np.abs(ref - g).max(1)
            d[used] = np.inf
            j = int(np.argmin(d))
            if d[j] <= atol:
                used[j] = True
                continue
        left_got.append(g)
    left_ref = [r for r, u in zip(ref, used) if not u]

    def excusable(x, others):
        if abs(float(x[4]) - min_conf) <= atol:
            return True
        return any(abs(_iou(x[:4].astype(np.float64), o[:4].astype(np.float64)) - max_iou) <= 1e-3 for o in others)

    bad = [x for x in left_got if not excusable(x, ref)] + [x for x in left_ref if not excusable(x, got)]
    n_left = len(left_got) + len(left_ref)
    if bad or n_left > max(2, 0.01 * max(len(got), len(ref))):
        raise AssertionError("%s detections differ: got %d, oracle %d, unmatched %d (%d not borderline)\n got=%s\n ref=%s" %
                             (what, len(got), len(ref), n_left, len(bad), np.array(left_got[:4]), np.array(left_ref[:4])))
    EXCUSED["frames"] += 1 if n_left else 0
    EXCUSED["detections"] += n_left
    return n_left  # number of excused (borderline) detections: callers log how often the excuse fires


# how often assert_dets_match excused a borderline decision in this test session (printed by conftest)
EXCUSED = {"frames": 0, "detections": 0}


# what the reference itself pins (integration_tests.rs:20-35) and whether this session could check it (printed by conftest)
REFERENCE_PINS = {}


def dets_from_ctypes(out, cnt, cap, i):
    """Frame i of a UfdDet array laid out [count][cap] -> [n,5] f32 without per-element ctypes access."""
    n = min(int(cnt[i]), cap)
    a = np.frombuffer(out, np.float32).reshape(-1, cap, 5)
    return a[i, :n].copy()


def oracle_many(fn, items, threads=16):
    """Runs the (GIL-releasing, ctypes) oracle over many frames on a few host threads."""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(fn, items))
