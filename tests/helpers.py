"""Shared helpers for the parity tests."""
import numpy as np


def dets_array(dets):
    """[(bbox, conf)] -> [n,5] f32"""
    return np.array([list(b) + [c] for b, c in dets], np.float32).reshape(-1, 5)


def assert_dets_match(got, ref, scores=None, min_conf=0.5, atol=1e-4, what=""):
    """Detection lists must agree: same count, same order, coordinates/confidence within atol.
    A count difference is tolerated only if a candidate sits within atol of the confidence
    threshold (fp32 rounding can flip a strict `>`); NMS IoU ties are not excused."""
    got, ref = np.asarray(got, np.float32).reshape(-1, 5), np.asarray(ref, np.float32).reshape(-1, 5)
    if got.shape == ref.shape:
        if got.size == 0 or np.abs(got - ref).max() <= atol:
            return
    if scores is not None:
        near = np.abs(np.asarray(scores)[:, 1] - min_conf) <= atol
        if near.any():
            # drop borderline candidates from both lists and compare the rest
            def strip(d):
                return d[np.abs(d[:, 4] - min_conf) > atol]
            g, r = strip(got), strip(ref)
            if g.shape == r.shape and (g.size == 0 or np.abs(g - r).max() <= atol):
                return
    raise AssertionError("%s detections differ: got %d, oracle %d\n got=%s\n ref=%s" %
                         (what, len(got), len(ref), got[:5], ref[:5]))
