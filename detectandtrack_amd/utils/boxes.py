"""Box / tube arithmetic on the host (NumPy) — mirror of reference lib/utils/boxes.py.

Boxes are [x1, y1, x2, y2] with the "+1" width convention; a tube of T frames is the 4T-vector of its per-frame
boxes, and per-class tube predictions are laid out class-major, then frame, then xyxy (reference :26-57).
"""
import numpy as np

from detectandtrack_amd.core.config import cfg


def split_tube_into_boxes(tube, T=None):
    """(N, 4T[+1]) or (N, 4T*num_classes) -> list of T per-frame (N, 4[*num_classes][+1]) arrays (:26-57)."""
    n = tube.shape[0]
    if tube.shape[1] % 4 == 0:
        tail = np.zeros((n, 0))
    elif (tube.shape[1] - 1) % 4 == 0:
        tail, tube = tube[:, (-1,)], tube[:, :-1]
    else:
        raise ValueError('Invalid tube dimensions {}'.format(tube.shape))
    T = T or tube.shape[-1] // 4
    if 4 * T == tube.shape[-1]:
        frames = [tube[..., 4 * t:4 * (t + 1)] for t in range(T)]
    else:
        assert tube.shape[-1] % (4 * T) == 0
        k = tube.shape[-1] // (4 * T)
        per_class = tube.reshape(n, k, T, 4)
        frames = [per_class[:, :, t, :].reshape(n, 4 * k) for t in range(T)]
    return [np.hstack((f, tail)) for f in frames], T


def _iou_matrix(boxes, query):
    """Pairwise IoU of (N,4) vs (K,4) float32 boxes with the exact evaluation order of the C that Cython emits
    for lib/utils/cython_bbox.pyx:16-57 (differences of float32 are float32; `+ 1` is a DOUBLE constant, so areas
    and the union are formed in double and rounded once to float32; the final product/quotient are float32)."""
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    q = np.ascontiguousarray(query, dtype=np.float32)
    f64, f32 = np.float64, np.float32
    bw = (b[:, 2] - b[:, 0]).astype(f64)[:, None] + 1.0
    bh = (b[:, 3] - b[:, 1]).astype(f64)[:, None] + 1.0
    qa = (((q[:, 2] - q[:, 0]).astype(f64) + 1.0) * ((q[:, 3] - q[:, 1]).astype(f64) + 1.0)).astype(f32)[None, :]
    iw = ((np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0])).astype(f64) + 1.0).astype(f32)
    ih = ((np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1])).astype(f64) + 1.0).astype(f32)
    inter = iw * ih
    ua = (bw * bh + qa.astype(f64) - inter.astype(f64)).astype(f32)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = inter / ua
    return np.where((iw > 0) & (ih > 0), iou, f32(0)).astype(f32)


def bbox_overlaps(boxes, query_boxes):
    """IoU, averaged over frames for tubes (:60-69)."""
    parts, _ = split_tube_into_boxes(boxes)
    qparts, _ = split_tube_into_boxes(query_boxes)
    return np.mean(np.stack([_iou_matrix(p, q) for p, q in zip(parts, qparts)]), axis=0)


def boxes_area(boxes):
    """Area; for tubes the mean over frames (:72-78)."""
    w = boxes[:, 2::4] - boxes[:, 0::4] + 1
    h = boxes[:, 3::4] - boxes[:, 1::4] + 1
    areas = np.mean(w * h, axis=1)
    assert np.all(areas >= 0), 'Negative areas founds'
    return areas


def bbox_transform(boxes, deltas, weights):
    """Apply (dx, dy, dw, dh)/weights to boxes; dw/dh clipped at cfg.BBOX_XFORM_CLIP (:141-183)."""
    if boxes.shape[1] > 4:
        return tube_transform(boxes, deltas, weights)
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    boxes = boxes.astype(deltas.dtype, copy=False)
    w = boxes[:, 2] - boxes[:, 0] + 1.0
    h = boxes[:, 3] - boxes[:, 1] + 1.0
    cx = boxes[:, 0] + 0.5 * w
    cy = boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw = np.minimum(deltas[:, 2::4] / ww, cfg.BBOX_XFORM_CLIP)
    dh = np.minimum(deltas[:, 3::4] / wh, cfg.BBOX_XFORM_CLIP)
    pcx = dx * w[:, np.newaxis] + cx[:, np.newaxis]
    pcy = dy * h[:, np.newaxis] + cy[:, np.newaxis]
    pw = np.exp(dw) * w[:, np.newaxis]
    ph = np.exp(dh) * h[:, np.newaxis]
    out = np.zeros(deltas.shape, dtype=deltas.dtype)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw
    out[:, 3::4] = pcy + 0.5 * ph
    return out


def tube_transform(boxes, deltas, weights):
    """Per-frame bbox_transform, re-interleaved class-major/frame/xyxy (:186-202)."""
    bparts, T = split_tube_into_boxes(boxes)
    dparts, _ = split_tube_into_boxes(deltas, T)
    tx = [bbox_transform(b, d, weights) for b, d in zip(bparts, dparts)]
    k = tx[0].shape[-1] // 4
    n = deltas.shape[0]
    res = np.zeros((n, k, T, 4), dtype=deltas.dtype)
    for t in range(T):
        res[:, :, t, :] = tx[t].reshape(n, k, 4)
    return res.reshape(deltas.shape)


def bbox_transform_inv(ex_rois, gt_rois, weights):
    """Regression targets from (proposal, ground truth) pairs (:205-239)."""
    if ex_rois.shape[1] > 4:
        assert ex_rois.shape[1] == gt_rois.shape[1]
        e, _ = split_tube_into_boxes(ex_rois)
        g, _ = split_tube_into_boxes(gt_rois)
        return np.concatenate([bbox_transform_inv(a, b, weights) for a, b in zip(e, g)], axis=1)
    ew = ex_rois[:, 2] - ex_rois[:, 0] + 1.0
    eh = ex_rois[:, 3] - ex_rois[:, 1] + 1.0
    gw = gt_rois[:, 2] - gt_rois[:, 0] + 1.0
    gh = gt_rois[:, 3] - gt_rois[:, 1] + 1.0
    wx, wy, ww, wh = weights
    dx = wx * ((gt_rois[:, 0] + 0.5 * gw) - (ex_rois[:, 0] + 0.5 * ew)) / ew
    dy = wy * ((gt_rois[:, 1] + 0.5 * gh) - (ex_rois[:, 1] + 0.5 * eh)) / eh
    return np.vstack((dx, dy, ww * np.log(gw / ew), wh * np.log(gh / eh))).transpose()


def clip_tiled_boxes(boxes, im_shape):
    """Clip every (x1, y1, x2, y2) group to [0, W-1] x [0, H-1]; im_shape = [H, W] (:243-253)."""
    for c, lim in ((0, im_shape[1]), (1, im_shape[0]), (2, im_shape[1]), (3, im_shape[0])):
        boxes[:, c::4] = np.maximum(np.minimum(boxes[:, c::4], lim - 1), 0)
    return boxes


def box_voting(top_dets, all_dets, thresh):
    """Refine `top_dets` [N, 5] by score-weighted averaging of the `all_dets` boxes that overlap each by >= thresh (:294-310,
    https://arxiv.org/abs/1505.01749)."""
    out = top_dets.copy()
    all_boxes, all_scores = all_dets[:, :4], all_dets[:, 4]
    overlaps = bbox_overlaps(top_dets[:, :4], all_boxes)
    for k in range(out.shape[0]):
        vote = np.where(overlaps[k] >= thresh)[0]
        out[k, :4] = np.average(all_boxes[vote, :], axis=0, weights=all_scores[vote])
    return out


def xywh_to_xyxy(boxes):
    boxes = np.asarray(boxes)
    if boxes.ndim == 1:
        return np.hstack((boxes[0:2], boxes[0:2] + boxes[2:4] - 1))
    return np.hstack((boxes[:, 0:2], boxes[:, 0:2] + boxes[:, 2:4] - 1))


def xyxy_to_xywh(boxes):
    boxes = np.asarray(boxes)
    return np.hstack((boxes[:, 0:2], boxes[:, 2:4] - boxes[:, 0:2] + 1))
