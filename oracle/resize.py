"""Oracle: OpenCV `cv2.resize` for float32 images, INTER_LINEAR and INTER_CUBIC (test infrastructure, see oracle/__init__.py).

PARITY STATUS: OpenCV itself is a third-party dependency absent from /root/reference (the reference environment pins
opencv 3.4.1, all_pkg_versions.txt:169) and is not installed here.  This file restates the portable C++ path of
`cv::resize` in that version (modules/imgproc/src/resize.cpp: `resize` set-up loop, `HResizeLinear` / `HResizeCubic`,
`VResizeLinear` / `VResizeCubic`, `interpolateCubic`) for CV_32F data; it is pinned by known answers derived with exact
rational arithmetic from the published kernel definitions (tests/test_oracle_golden.py), not by the library's output.

Reference call sites this oracle stands in for:
  * lib/utils/keypoints.py:129-131  `cv2.resize(maps[i], (w, h), interpolation=cv2.INTER_CUBIC)`  (dsize given)
  * lib/utils/blob.py:86-87         `cv2.resize(im, None, None, fx=s, fy=s, interpolation=cv2.INTER_LINEAR)` (scale given)

Algorithm (float32 throughout, every product and sum rounded to float32, summed left to right, no fma):
  * output size: `dsize` if given, else (round_half_even(W*fx), round_half_even(H*fy)); the sampling step per axis is
    scale = 1 / inv_scale (a double), inv_scale = fx when the scale was given and dst/src when dsize was given;
  * destination index d samples source coordinate f = float32((d + 0.5) * scale - 0.5); s = floor(f); t = f - s (float32);
  * INTER_LINEAR: taps s, s+1 with weights (1 - t, t).  Columns: s < 0 -> (s, t) = (0, 0); s >= n-1 -> (s, t) = (n-1, 0) (the
    x set-up loop; HResizeLinear then reads the border pixel with weight 1).  Rows: (s, t) are kept and the two row
    indices are clamped to [0, n-1] when fetched (resizeGeneric_Invoker), so a border row is blended with itself;
  * INTER_CUBIC: taps s-1 .. s+2, indices clamped to [0, n-1] (border replication), weights from the Keys kernel with
    A = -0.75 in OpenCV's Horner form, w3 = 1 - w0 - w1 - w2;
  * the horizontal pass runs first (every needed source row is resampled to the output width), then the vertical pass
    combines 2 / 4 resampled rows.
"""
import numpy as np

F32 = np.float32


def _out_len(n, inv_scale):
    return int(np.rint(n * inv_scale))          # saturate_cast<int>(double) == cvRound: round half to even


def _coords(n_src, n_dst, inv_scale):
    scale = 1.0 / inv_scale                     # double
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(F32)
    s = np.floor(f).astype(np.int64)
    t = (f - s.astype(F32)).astype(F32)
    return s, t


def _linear_taps(n_src, n_dst, inv_scale, horizontal):
    s, t = _coords(n_src, n_dst, inv_scale)
    if horizontal:      # the x set-up loop snaps out-of-range columns onto the border pixel with weight (1, 0) ...
        lo = s < 0
        s[lo], t[lo] = 0, F32(0)
        hi = s >= n_src - 1
        s[hi], t[hi] = n_src - 1, F32(0)
    # ... the y loop does not: it keeps (1 - t, t) and clamps the two ROW indices when the rows are fetched
    idx = np.stack([np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1)], axis=0)
    w = np.stack([(F32(1) - t).astype(F32), t], axis=0)
    return idx, w


def cubic_weights(t):
    """interpolateCubic: the Keys kernel, A = -0.75, evaluated at distances t+1, t, 1-t, 2-t (float32 Horner form)."""
    t = np.asarray(t, dtype=F32)
    A = F32(-0.75)
    one = F32(1)
    w0 = ((A * (t + one) - F32(5) * A) * (t + one) + F32(8) * A) * (t + one) - F32(4) * A
    w1 = ((A + F32(2)) * t - (A + F32(3))) * t * t + one
    u = one - t
    w2 = ((A + F32(2)) * u - (A + F32(3))) * u * u + one
    w3 = one - w0 - w1 - w2
    return np.stack([w0, w1, w2, w3], axis=0).astype(F32)


def _cubic_taps(n_src, n_dst, inv_scale):
    s, t = _coords(n_src, n_dst, inv_scale)
    idx = np.stack([np.clip(s - 1 + k, 0, n_src - 1) for k in range(4)], axis=0)
    return idx, cubic_weights(t)


def _separable(im, taps_x, taps_y):
    """Horizontal pass then vertical pass; products and partial sums in float32, left to right."""
    im = np.asarray(im, dtype=F32)
    extra = (1,) * (im.ndim - 2)
    ix, wx = taps_x
    rows = None
    for k in range(ix.shape[0]):                               # D[dx] = S[x0]*a0 + S[x1]*a1 (+ ...)
        term = (np.take(im, ix[k], axis=1) * wx[k].reshape((1, -1) + extra)).astype(F32)
        rows = term if rows is None else (rows + term).astype(F32)
    iy, wy = taps_y
    out = None
    for k in range(iy.shape[0]):                               # dst[x] = S0[x]*b0 + S1[x]*b1 (+ ...)
        term = (np.take(rows, iy[k], axis=0) * wy[k].reshape((-1, 1) + extra)).astype(F32)
        out = term if out is None else (out + term).astype(F32)
    return out


def _geometry(im, dsize, fx, fy):
    h, w = im.shape[:2]
    if dsize is not None:
        out_w, out_h = int(dsize[0]), int(dsize[1])
        return out_w, out_h, float(out_w) / w, float(out_h) / h
    return _out_len(w, fx), _out_len(h, fy), float(fx), float(fy)


def resize_linear(im, dsize=None, fx=None, fy=None):
    """cv2.resize(im, dsize or None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR) for float32 (H, W[, C]) images."""
    im = np.asarray(im, dtype=F32)
    out_w, out_h, isx, isy = _geometry(im, dsize, fx, fy)
    return _separable(im, _linear_taps(im.shape[1], out_w, isx, True), _linear_taps(im.shape[0], out_h, isy, False))


def resize_cubic(im, dsize=None, fx=None, fy=None):
    """cv2.resize(im, dsize or None, fx=fx, fy=fy, interpolation=cv2.INTER_CUBIC) for float32 (H, W[, C]) images."""
    im = np.asarray(im, dtype=F32)
    out_w, out_h, isx, isy = _geometry(im, dsize, fx, fy)
    return _separable(im, _cubic_taps(im.shape[1], out_w, isx), _cubic_taps(im.shape[0], out_h, isy))


# ---- the reference's heatmap decoding on top of it (lib/utils/keypoints.py:94-149, :210-216) ---------------------------
def scores_to_probs(scores):
    """Spatial softmax per channel of a (C, H, W) map (lib/utils/keypoints.py:210-216)."""
    scores = scores.copy()
    for c in range(scores.shape[0]):
        temp = scores[c, :, :]
        max_score = temp.max()
        temp = np.exp(temp - max_score) / np.sum(np.exp(temp - max_score))
        scores[c, :, :] = temp
    return scores


def heatmaps_to_keypoints(maps, rois, min_size=0):
    """maps (R, K, M, M) float32 logits, rois (R, 4) -> (R, 4, K) rows (x, y, logit, prob): lib/utils/keypoints.py:94-149
    with cv2.resize replaced by resize_cubic above."""
    offset_x = rois[:, 0]
    offset_y = rois[:, 1]
    widths = np.maximum(rois[:, 2] - rois[:, 0], 1)
    heights = np.maximum(rois[:, 3] - rois[:, 1], 1)
    widths_ceil = np.ceil(widths)
    heights_ceil = np.ceil(heights)
    maps = np.transpose(maps, [0, 2, 3, 1])                    # NCHW -> NHWC (:111)
    num_kps = maps.shape[3]
    xy_preds = np.zeros((len(rois), 4, num_kps), dtype=F32)
    for i in range(len(rois)):
        if min_size > 0:
            mw = int(np.maximum(widths_ceil[i], min_size))
            mh = int(np.maximum(heights_ceil[i], min_size))
        else:
            mw, mh = int(widths_ceil[i]), int(heights_ceil[i])
        width_correction = widths[i] / mw
        height_correction = heights[i] / mh
        roi_map = np.transpose(resize_cubic(maps[i], dsize=(mw, mh)), [2, 0, 1])
        roi_map_probs = scores_to_probs(roi_map.copy())
        w = roi_map.shape[2]
        for k in range(num_kps):
            pos = roi_map[k, :, :].argmax()
            x_int = pos % w
            y_int = (pos - x_int) // w
            assert roi_map_probs[k, y_int, x_int] == roi_map_probs[k, :, :].max()
            x = (x_int + 0.5) * width_correction
            y = (y_int + 0.5) * height_correction
            xy_preds[i, 0, k] = x + offset_x[i]
            xy_preds[i, 1, k] = y + offset_y[i]
            xy_preds[i, 2, k] = roi_map[k, y_int, x_int]
            xy_preds[i, 3, k] = roi_map_probs[k, y_int, x_int]
    return xy_preds
