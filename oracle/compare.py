"""How two detection lists of one frame are compared -- TEST INFRASTRUCTURE (tests/helpers.py and bench.py's `verified` block).

north_star's bar is 1e-3 on boxes and confidences.  fp32 rounding can do two things to a list that no tolerance on the
numbers covers: swap two detections whose confidences differ by less than the rounding, and flip a decision that sat on its
threshold (the strict `conf > min_conf` of nn.rs:121-128, the strict `iou > max_iou` of nn.rs:209-214).  So the lists are
matched as SETS, and a detection without a partner is excused only when the decision that produced it is provably
borderline at fp32 resolution.
"""
import numpy as np


def iou64(a, b):
    """nn.rs:227-243 in float64 (only used to recognise borderline NMS decisions)"""
    def area(x):
        w, h = x[3] - x[1], x[2] - x[0]
        return 0.0 if (w < 0 or h < 0) else w * h
    o = [max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])]
    ov = area(o)
    return ov / (area(a) + area(b) - ov + 1e-7)


def match_detections(got, ref, min_conf=0.5, max_iou=0.5, atol=1e-4):
    """-> dict(equal, max_err, left_got, left_ref, not_borderline)
    equal           same length and element-wise within atol, order included
    max_err         largest |difference| over the matched pairs (set matching, nearest partner within atol)
    left_got/ref    detections without a partner in the other list
    not_borderline  those of them whose decision was NOT within atol of the confidence threshold nor within 1e-3 of max_iou
                    against some detection of the other list: real disagreements"""
    got, ref = np.asarray(got, np.float32).reshape(-1, 5), np.asarray(ref, np.float32).reshape(-1, 5)
    if got.shape == ref.shape and (got.size == 0 or np.abs(got - ref).max() <= atol):
        return dict(equal=True, max_err=float(np.abs(got - ref).max()) if got.size else 0.0, left_got=[], left_ref=[], not_borderline=[])
    used = np.zeros(len(ref), bool)
    left_got, max_err = [], 0.0
    for g in got:
        if len(ref):
            d = np.abs(ref - g).max(1)
            d[used] = np.inf
            j = int(np.argmin(d))
            if d[j] <= atol:
                used[j] = True
                max_err = max(max_err, float(d[j]))
                continue
        left_got.append(g)
    left_ref = [r for r, u in zip(ref, used) if not u]

    def excusable(x, others):
        if abs(float(x[4]) - min_conf) <= atol:
            return True
        return any(abs(iou64(x[:4].astype(np.float64), o[:4].astype(np.float64)) - max_iou) <= 1e-3 for o in others)

    bad = [x for x in left_got if not excusable(x, ref)] + [x for x in left_ref if not excusable(x, got)]
    return dict(equal=False, max_err=max_err, left_got=left_got, left_ref=left_ref, not_borderline=bad)
