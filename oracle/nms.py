"""Oracle: greedy NMS for boxes and tubes (test infrastructure, see oracle/__init__.py).

Follows reference lib/utils/cython_nms.pyx:37-87 (boxes, suppress at IoU >= thr,
returns ASCENDING original indices) and lib/nms/py_cpu_nms_tubes.py:17-53
(tubes, per-frame IoU averaged over T, keep while mean <= thr, returns indices
in SCORE order), dispatched as lib/core/nms_wrapper.py:49-57.

Ordering note: the reference sorts with NumPy's unstable `argsort()[::-1]`, so
its order among exactly-tied scores is an implementation accident.  The oracle
(and the HIP kernels) define ties as "lower original index first"; on tie-free
inputs this is identical to the reference (pinned against the compiled
reference in oracle/_ref, see tests/test_oracle_ref.py).
"""
import numpy as np


def _order_desc(scores):
    return np.argsort(-scores.astype(np.float32), kind='stable')


def nms_boxes(dets, thresh):
    """dets: (n, 5) float32 [x1 y1 x2 y2 score]; returns int64 ascending keep."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    thresh = np.float32(thresh)
    one = np.float32(1)
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    areas = (x2 - x1 + one) * (y2 - y1 + one)
    order = _order_desc(dets[:, 4])
    suppressed = np.zeros(n, dtype=bool)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1 + one)
        h = np.maximum(np.float32(0), yy2 - yy1 + one)
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr >= thresh]] = True
    return np.where(~suppressed)[0]


def nms_tubes(dets, thresh):
    """dets: (n, 4T+1) float32; returns list of indices in score order."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    T = (dets.shape[1] - 1) // 4
    one = np.float32(1)
    thresh = np.float32(thresh)
    areas = [(dets[:, 4 * t + 2] - dets[:, 4 * t + 0] + one) *
             (dets[:, 4 * t + 3] - dets[:, 4 * t + 1] + one) for t in range(T)]
    order = _order_desc(dets[:, -1])
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        ovT = np.zeros(rest.shape[0], dtype=np.float32)
        for t in range(T):
            xx1 = np.maximum(dets[i, 4 * t + 0], dets[rest, 4 * t + 0])
            yy1 = np.maximum(dets[i, 4 * t + 1], dets[rest, 4 * t + 1])
            xx2 = np.minimum(dets[i, 4 * t + 2], dets[rest, 4 * t + 2])
            yy2 = np.minimum(dets[i, 4 * t + 3], dets[rest, 4 * t + 3])
            w = np.maximum(np.float32(0), xx2 - xx1 + one)
            h = np.maximum(np.float32(0), yy2 - yy1 + one)
            inter = w * h
            with np.errstate(divide='ignore', invalid='ignore'):
                ovT = ovT + inter / (areas[t][i] + areas[t][rest] - inter)
        ovT = ovT / np.float32(T)
        order = rest[ovT <= thresh]
    return keep


def nms(dets, thresh):
    """core/nms_wrapper.py:49-57."""
    if dets.shape[0] == 0:
        return []
    if dets.shape[1] > 5:
        return nms_tubes(dets, thresh)
    return nms_boxes(dets, thresh)


def gpu_nms_presorted(boxes, thresh):
    """The reference's `_nms` C symbol (lib/nms/gpu_nms.hpp:3-9, implementation lib/nms/nms_kernel.cu:25-150): rows [n, 5]
    PRE-SORTED by score are visited in the given order; row j > i is removed by a kept row i when
    devIoU(i, j) = inter / (Sa + Sb - inter) > thresh (STRICT, :71; widths/heights +1, :28-35); returns the kept row
    positions in order (:128-141).  Pinned by hand-checkable cases in tests/test_oracle_golden.py."""
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    n = b.shape[0]
    thresh = np.float32(thresh)
    one = np.float32(1)
    area = (b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one)
    removed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        j = np.arange(i + 1, n)
        w = np.maximum(np.minimum(b[i, 2], b[j, 2]) - np.maximum(b[i, 0], b[j, 0]) + one, np.float32(0))
        h = np.maximum(np.minimum(b[i, 3], b[j, 3]) - np.maximum(b[i, 1], b[j, 1]) + one, np.float32(0))
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            iou = inter / (area[i] + area[j] - inter)
        removed[j[iou > thresh]] = True
    return np.asarray(keep, dtype=np.int32)
