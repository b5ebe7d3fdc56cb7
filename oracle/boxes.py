"""Oracle: box / tube arithmetic (test infrastructure, see oracle/__init__.py).

Follows reference lib/utils/boxes.py and lib/utils/cython_bbox.pyx.
"""
import numpy as np

BBOX_XFORM_CLIP = np.log(1000. / 16.)  # core/config.py:672


def split_tube_into_boxes(tube, T=None):
    """utils/boxes.py:26-57."""
    N = tube.shape[0]
    if tube.shape[1] % 4 == 0:
        scores = np.zeros((N, 0))
    elif (tube.shape[1] - 1) % 4 == 0:
        scores = tube[:, (-1,)]
        tube = tube[:, :-1]
    else:
        raise ValueError('Invalid tube dimensions {}'.format(tube.shape))
    T = T or tube.shape[-1] // 4
    parts = []
    if 4 * T != tube.shape[-1]:
        assert tube.shape[-1] % (4 * T) == 0
        ncls = tube.shape[-1] // (4 * T)
        for t in range(T):
            rep = np.zeros((N, 4 * ncls))
            for c in range(ncls):
                rep[:, c * 4:(c + 1) * 4] = tube[:, c * 4 * T:(c + 1) * 4 * T][:, t * 4:(t + 1) * 4]
            parts.append(rep)
    else:
        for t in range(T):
            parts.append(tube[..., t * 4:(t + 1) * 4])
    return [np.hstack((p, scores)) for p in parts], T


def bbox_overlaps_2d(boxes, query_boxes):
    """Pairwise IoU, float32, +1 convention.  utils/cython_bbox.pyx:16-57.

    Written as the same scalar float32 expression sequence as the Cython loop
    (numpy float32 scalars round after every op exactly as C floats do).
    """
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    q = np.ascontiguousarray(query_boxes, dtype=np.float32)
    N, K = boxes.shape[0], q.shape[0]
    out = np.zeros((N, K), dtype=np.float32)
    f64, f32 = np.float64, np.float32
    # Evaluation order/types of the C that Cython emits for the .pyx (checked against the
    # compiled reference in oracle/_ref): `a - b` of two float32 is a float32 op, the literal
    # `+ 1` becomes the DOUBLE constant 1.0 (so the rest of that expression is double), and
    # every assignment to a DTYPE_t variable rounds once to float32.
    bw = (boxes[:, 2] - boxes[:, 0]).astype(f64) + 1.0
    bh = (boxes[:, 3] - boxes[:, 1]).astype(f64) + 1.0
    for k in range(K):
        qarea = f32((f64(q[k, 2] - q[k, 0]) + 1.0) * (f64(q[k, 3] - q[k, 1]) + 1.0))
        iw = ((np.minimum(boxes[:, 2], q[k, 2]) - np.maximum(boxes[:, 0], q[k, 0])).astype(f64) + 1.0).astype(f32)
        ih = ((np.minimum(boxes[:, 3], q[k, 3]) - np.maximum(boxes[:, 1], q[k, 1])).astype(f64) + 1.0).astype(f32)
        inter = iw * ih  # float32 product
        ua = (bw * bh + f64(qarea) - inter.astype(f64)).astype(f32)
        ok = (iw > 0) & (ih > 0)
        with np.errstate(divide='ignore', invalid='ignore'):
            v = inter / ua
        out[ok, k] = v[ok]
    return out


def bbox_overlaps(boxes, query_boxes):
    """Tube-mean IoU.  utils/boxes.py:60-69."""
    parts, _ = split_tube_into_boxes(boxes)
    qparts, _ = split_tube_into_boxes(query_boxes)
    return np.mean(np.stack([
        bbox_overlaps_2d(p.astype(np.float32, copy=False), q.astype(np.float32, copy=False))
        for p, q in zip(parts, qparts)]), axis=0)


def boxes_area(boxes):
    """utils/boxes.py:72-78 (tube: mean over frames)."""
    w = boxes[:, 2::4] - boxes[:, 0::4] + 1
    h = boxes[:, 3::4] - boxes[:, 1::4] + 1
    return np.mean(w * h, axis=1)


def bbox_transform(boxes, deltas, weights=(1.0, 1.0, 1.0, 1.0)):
    """Apply regression deltas.  utils/boxes.py:141-183 (tubes: :186-202)."""
    if boxes.shape[1] > 4:
        return tube_transform(boxes, deltas, weights)
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    boxes = boxes.astype(deltas.dtype, copy=False)
    widths = boxes[:, 2] - boxes[:, 0] + 1.0
    heights = boxes[:, 3] - boxes[:, 1] + 1.0
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = np.minimum(deltas[:, 2::4] / ww, BBOX_XFORM_CLIP)
    dh = np.minimum(deltas[:, 3::4] / wh, BBOX_XFORM_CLIP)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = np.exp(dw) * widths[:, None]
    ph = np.exp(dh) * heights[:, None]
    out = np.zeros(deltas.shape, dtype=deltas.dtype)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw
    out[:, 3::4] = pcy + 0.5 * ph
    return out


def tube_transform(boxes, deltas, weights):
    """utils/boxes.py:186-202."""
    bparts, T = split_tube_into_boxes(boxes)
    dparts, _ = split_tube_into_boxes(deltas, T)
    tx = [bbox_transform(b, d, weights) for b, d in zip(bparts, dparts)]
    ncls = tx[0].shape[-1] // 4
    res = np.zeros(deltas.shape, dtype=deltas.dtype)
    for c in range(ncls):
        for t in range(T):
            res[:, c * 4 * T:(c + 1) * 4 * T][:, t * 4:(t + 1) * 4] = tx[t][:, c * 4:(c + 1) * 4]
    return res


def bbox_transform_inv(ex, gt, weights=(1.0, 1.0, 1.0, 1.0)):
    """utils/boxes.py:205-229 (tubes: :232-239)."""
    if ex.shape[1] > 4:
        eparts, _ = split_tube_into_boxes(ex)
        gparts, _ = split_tube_into_boxes(gt)
        return np.concatenate([bbox_transform_inv(e, g, weights)
                               for e, g in zip(eparts, gparts)], axis=1)
    ew = ex[:, 2] - ex[:, 0] + 1.0
    eh = ex[:, 3] - ex[:, 1] + 1.0
    ecx = ex[:, 0] + 0.5 * ew
    ecy = ex[:, 1] + 0.5 * eh
    gw = gt[:, 2] - gt[:, 0] + 1.0
    gh = gt[:, 3] - gt[:, 1] + 1.0
    gcx = gt[:, 0] + 0.5 * gw
    gcy = gt[:, 1] + 0.5 * gh
    wx, wy, ww, wh = weights
    return np.vstack((wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh,
                      ww * np.log(gw / ew), wh * np.log(gh / eh))).transpose()


def clip_tiled_boxes(boxes, im_shape):
    """utils/boxes.py:243-253."""
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes
