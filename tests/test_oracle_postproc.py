"""Oracle pinning, rows A7-A10: known-answer cases authored from the first-party reference code
(infer_server/src/nn.rs:109-140 postproc, :198-224 NMS, :227-243 iou, :251-260 bbox_area)."""
import numpy as np


def mk(confs, boxes):
    c = np.asarray(confs, np.float32)
    return np.stack([1 - c, c], 1).astype(np.float32), np.asarray(boxes, np.float32)


def test_bbox_area_and_iou(oracle_lib):
    assert oracle_lib.bbox_area([0, 0, 2, 3]) == 6.0
    assert oracle_lib.bbox_area([0.5, 0.5, 0.4, 0.9]) == 0.0  # x_br < x_tl: ill-defined -> 0 (nn.rs:254-257)
    assert oracle_lib.bbox_area([0.5, 0.5, 0.9, 0.4]) == 0.0
    a, b = [0, 0, 1, 1], [0.5, 0, 1.5, 1]
    f = np.float32
    expect = f(0.5) / f(f(f(1) + f(1)) - f(0.5) + f(1e-7))
    assert oracle_lib.iou(a, b) == float(expect)
    assert oracle_lib.iou(a, [2, 2, 3, 3]) == 0.0  # disjoint: overlap box ill-defined -> area 0
    assert oracle_lib.iou([0, 0, 0, 0], [0, 0, 0, 0]) == 0.0  # EPS avoids 0/0 (nn.rs:18,242)
    assert abs(oracle_lib.iou(a, a) - 1.0) < 1e-6


def test_threshold_is_strict_and_nan_dropped(oracle_lib):
    s, b = mk([0.5, 0.50001, 0.49, np.nan, 0.9], [[0, 0, .1, .1], [.2, .2, .3, .3], [.4, .4, .5, .5], [.6, .6, .7, .7],
                                                   [.8, .8, .9, .9]])
    d = oracle_lib.postproc(s, b, 0.5, 0.5)
    assert len(d) == 2 and np.allclose(d[:, 4], [0.9, 0.50001])  # descending confidence (nn.rs:107-108)


def test_nms_suppresses_strictly_above_max_iou(oracle_lib):
    # iou(a, b) where b is a shifted copy: 1/3 overlap-over-union at shift .5
    boxes = [[0, 0, 1, 1], [0.5, 0, 1.5, 1], [0, 0, 1, 1.0000001]]
    s, b = mk([0.9, 0.8, 0.7], boxes)
    d = oracle_lib.postproc(s, b, 0.5, 0.5)
    assert len(d) == 2 and np.allclose(d[:, 4], [0.9, 0.8])  # third box iou ~1 > .5 -> suppressed
    i = oracle_lib.iou(boxes[0], boxes[1])
    d = oracle_lib.postproc(s[:2], b[:2], 0.5, i)  # iou == max_iou is NOT suppressed (strict >)
    assert len(d) == 2
    d = oracle_lib.postproc(s[:2], b[:2], 0.5, np.nextafter(np.float32(i), np.float32(0)))
    assert len(d) == 1


def test_tie_order_higher_index_first(oracle_lib):
    # equal confidences: stable ascending sort + pop() from the back => higher prior index first
    boxes = [[0, 0, 1, 1], [0.05, 0, 1.05, 1], [3, 3, 4, 4]]
    s, b = mk([0.75, 0.75, 0.75], boxes)
    d = oracle_lib.postproc(s, b, 0.5, 0.5)
    assert len(d) == 2
    assert np.allclose(d[0, :4], boxes[2]) and np.allclose(d[1, :4], boxes[1])  # index 1 beats index 0


def test_degenerate_boxes_never_suppress(oracle_lib):
    boxes = [[0.5, 0.5, 0.4, 0.4], [0.5, 0.5, 0.4, 0.4], [0.1, 0.1, 0.2, 0.2]]
    s, b = mk([0.9, 0.8, 0.7], boxes)
    assert len(oracle_lib.postproc(s, b, 0.5, 0.5)) == 3  # zero areas -> iou 0


def test_empty_and_all(oracle_lib):
    s, b = mk([0.1, 0.2], [[0, 0, 1, 1], [0, 0, 1, 1]])
    assert oracle_lib.postproc(s, b, 0.5, 0.5).shape == (0, 5)
    n = 300
    c = np.linspace(0.51, 0.99, n).astype(np.float32)
    bx = np.stack([np.arange(n) * 2.0, np.zeros(n), np.arange(n) * 2.0 + 1, np.ones(n)], 1).astype(np.float32)
    s, b = mk(c, bx)
    d = oracle_lib.postproc(s, b, 0.5, 0.5)
    assert len(d) == n and (np.diff(d[:, 4]) <= 0).all()


def test_greedy_matches_python_reference_randomised(oracle_lib):
    rng = np.random.default_rng(0)
    for _ in range(5):
        n = 400
        conf = rng.random(n).astype(np.float32)
        c = rng.random((n, 2)).astype(np.float32)
        sz = (rng.random((n, 2)) * 0.3 + 0.02).astype(np.float32)
        boxes = np.concatenate([c - sz / 2, c + sz / 2], 1).astype(np.float32)
        s = np.stack([1 - conf, conf], 1).astype(np.float32)
        order = sorted([i for i in range(n) if conf[i] > 0.5], key=lambda i: (conf[i], i))  # stable ascending
        sel = []
        while order:
            i = order.pop()
            if all(not (oracle_lib.iou(boxes[i], boxes[j]) > 0.5) for j in sel):
                sel.append(i)
        d = oracle_lib.postproc(s, boxes, 0.5, 0.5)
        assert len(d) == len(sel) and np.array_equal(d[:, :4], boxes[sel]) and np.array_equal(d[:, 4], conf[sel])
