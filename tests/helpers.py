"""Shared helpers for the parity tests."""
import numpy as np


def dets_array(dets):
    """[(bbox, conf)] -> [n,5] f32"""
    return np.array([list(b) + [c] for b, c in dets], np.float32).reshape(-1, 5)


def assert_dets_match(got, ref, scores=None, min_conf=0.5, max_iou=0.5, atol=1e-4, what=""):
    """Detection lists must agree within `atol` (north_star bar: 1e-3).

    1. Same length and element-wise close (order included): pass.
    2. Otherwise the lists are matched as sets (fp32 rounding can swap two detections whose
       confidences differ by < atol).  A detection without a partner is excused only when the
       decision that produced it is provably borderline at fp32 resolution:
         - its confidence is within atol of the threshold (strict `>` flipped), or
         - its IoU with some detection of the other list is within 1e-3 of max_iou (NMS flipped),
       and such leftovers must stay below 1 % of the list.  (The rule itself: oracle/compare.py, shared with bench.py.)"""
    from oracle.compare import match_detections

    got, ref = np.asarray(got, np.float32).reshape(-1, 5), np.asarray(ref, np.float32).reshape(-1, 5)
    r = match_detections(got, ref, min_conf, max_iou, atol)
    if r["equal"]:
        return 0
    left_got, left_ref, bad = r["left_got"], r["left_ref"], r["not_borderline"]
    n_left = len(left_got) + len(left_ref)
    if bad or n_left > max(2, 0.01 * max(len(got), len(ref))):
        raise AssertionError("%s detections differ: got %d, oracle %d, unmatched %d (%d not borderline)\n got=%s\n ref=%s" %
                             (what, len(got), len(ref), n_left, len(bad), np.array(left_got[:4]), np.array(left_ref[:4])))
    EXCUSED["frames"] += 1 if n_left else 0
    EXCUSED["detections"] += n_left
    return n_left  # number of excused (borderline) detections: callers log how often the excuse fires


# how often assert_dets_match excused a borderline decision in this test session (printed by conftest)
EXCUSED = {"frames": 0, "detections": 0}


# things a GPU session found out about its box that are worth a line in the summary (printed by conftest)
SESSION_NOTES = {}


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
