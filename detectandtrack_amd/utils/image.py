"""Host image helpers (no OpenCV in this environment): `cv2.resize` INTER_LINEAR / INTER_CUBIC for float32 images, and
the B*T -> T-axis move of reference lib/utils/image.py:82-93.

Resampling follows OpenCV's float32 path (the reference environment pins opencv 3.4.1): destination index d reads source
coordinate f = float32((d + 0.5) * step - 0.5), step = 1 / inv_scale in double, inv_scale = the caller's fx/fy when the scale
is given (lib/utils/blob.py:86-87) or dst/src when the size is given (lib/utils/keypoints.py:129-131); the horizontal pass runs
first, then the vertical one; cubic weights are the Keys kernel with a = -0.75 and replicated borders.  Checked against the
independent restatement in oracle/resize.py (tests/test_host_cpu.py)."""
import numpy as np


def _src_coords(n_dst, inv_scale):
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * (1.0 / inv_scale) - 0.5).astype(np.float32)
    i0 = np.floor(f)
    return i0.astype(np.int64), f - i0.astype(np.float32)


def _sizes(im, out_w, out_h, fx, fy):
    h, w = im.shape[:2]
    if out_w is None:
        return int(np.rint(w * fx)), int(np.rint(h * fy)), float(fx), float(fy)      # cvRound: half to even
    return int(out_w), int(out_h), float(out_w) / w, float(out_h) / h


def resize_bilinear(im, out_w=None, out_h=None, fx=None, fy=None):
    """im (H, W[, C]) float32 -> (out_h, out_w[, C]); cv2.INTER_LINEAR.  Either the output size or the scale factors.
    One horizontal pass over every SOURCE row, then the vertical blend of the gathered rows."""
    im = np.asarray(im, dtype=np.float32)
    h, w = im.shape[:2]
    out_w, out_h, isx, isy = _sizes(im, out_w, out_h, fx, fy)
    x0, tx = _src_coords(out_w, isx)
    y0, ty = _src_coords(out_h, isy)
    # columns outside the image snap onto the border pixel (weights 1, 0); rows keep their weights and clamp the indices
    tx = np.where((x0 < 0) | (x0 >= w - 1), np.float32(0), tx).astype(np.float32)
    x0 = np.clip(x0, 0, w - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    y1 = np.clip(y0 + 1, 0, h - 1)
    y0 = np.clip(y0, 0, h - 1)
    tail = (1,) * (im.ndim - 2)
    tx = tx.reshape((1, -1) + tail)
    ty = ty.reshape((-1, 1) + tail)
    rows = np.take(im, x0, axis=1)
    rows *= (np.float32(1) - tx)
    right = np.take(im, x1, axis=1)
    right *= tx
    rows += right                                   # (h, out_w[, C])
    out = np.take(rows, y0, axis=0)
    out *= (np.float32(1) - ty)
    below = np.take(rows, y1, axis=0)
    below *= ty
    out += below
    return out


def _keys_weights(t):
    """Four weights of the a = -0.75 bicubic kernel at fractional offset t (distances t+1, t, 1-t, 2-t)."""
    a = np.float32(-0.75)
    one = np.float32(1)
    p, q = t + one, one - t
    w0 = ((a * p - np.float32(5) * a) * p + np.float32(8) * a) * p - np.float32(4) * a
    w1 = ((a + np.float32(2)) * t - (a + np.float32(3))) * t * t + one
    w2 = ((a + np.float32(2)) * q - (a + np.float32(3))) * q * q + one
    return [w0, w1, w2, one - w0 - w1 - w2]


def resize_bicubic(im, out_w, out_h):
    """im (H, W[, C]) float32 -> (out_h, out_w[, C]); cv2.INTER_CUBIC (a = -0.75, replicated borders)."""
    im = np.asarray(im, dtype=np.float32)
    h, w = im.shape[:2]
    x0, tx = _src_coords(out_w, float(out_w) / w)
    y0, ty = _src_coords(out_h, float(out_h) / h)
    tail = (1,) * (im.ndim - 2)
    wx, wy = _keys_weights(tx), _keys_weights(ty)
    rows = np.take(im, np.clip(x0 - 1, 0, w - 1), axis=1) * wx[0].reshape((1, -1) + tail)
    for k in range(1, 4):
        rows += np.take(im, np.clip(x0 - 1 + k, 0, w - 1), axis=1) * wx[k].reshape((1, -1) + tail)
    out = np.take(rows, np.clip(y0 - 1, 0, h - 1), axis=0) * wy[0].reshape((-1, 1) + tail)
    for k in range(1, 4):
        out += np.take(rows, np.clip(y0 - 1 + k, 0, h - 1), axis=0) * wy[k].reshape((-1, 1) + tail)
    return out


def move_batch_to_time(blob, num_frames):
    """(B*T, C, H, W) -> (B, C, T, H, W) (reference :82-93)."""
    bt, c, h, w = blob.shape
    assert bt % num_frames == 0
    return np.ascontiguousarray(blob.reshape(bt // num_frames, num_frames, c, h, w).transpose(0, 2, 1, 3, 4))
