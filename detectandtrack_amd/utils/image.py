"""Host image helpers (no OpenCV in this environment): bilinear / bicubic resize with OpenCV's pixel-centre
convention (src = (dst + 0.5) * scale - 0.5, replicated borders) and the B*T -> T-axis move of reference
lib/utils/image.py:82-93."""
import numpy as np


def _axis_resample(src_len, dst_len):
    scale = float(src_len) / float(dst_len)
    s = (np.arange(dst_len, dtype=np.float64) + 0.5) * scale - 0.5
    i0 = np.floor(s).astype(np.int64)
    return i0, (s - i0)


def resize_bilinear(im, out_w, out_h):
    """im (H, W[, C]) float32 -> (out_h, out_w[, C]); cv2.INTER_LINEAR semantics.  Horizontal pass over every SOURCE row once,
    then the vertical blend of the gathered rows (same arithmetic per output as blending the four neighbours, 6x less work)."""
    im = np.asarray(im, dtype=np.float32)
    h, w = im.shape[:2]
    y0, fy = _axis_resample(h, out_h)
    x0, fx = _axis_resample(w, out_w)
    y1, x1 = np.clip(y0 + 1, 0, h - 1), np.clip(x0 + 1, 0, w - 1)
    y0, x0 = np.clip(y0, 0, h - 1), np.clip(x0, 0, w - 1)
    fy = fy.astype(np.float32).reshape((-1, 1) + (1,) * (im.ndim - 2))
    fx = fx.astype(np.float32).reshape((1, -1) + (1,) * (im.ndim - 2))
    rows = np.take(im, x0, axis=1)
    rows *= (1 - fx)
    right = np.take(im, x1, axis=1)
    right *= fx
    rows += right                                   # (h, out_w[, C])
    out = np.take(rows, y0, axis=0)
    out *= (1 - fy)
    below = np.take(rows, y1, axis=0)
    below *= fy
    out += below
    return out


def _cubic_coeffs(t, a=-0.75):
    t = t.astype(np.float32)
    c0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    c1 = ((a + 2) * t - (a + 3)) * t * t + 1
    c2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    return np.stack([c0, c1, c2, 1.0 - c0 - c1 - c2], axis=0).astype(np.float32)


def resize_bicubic(im, out_w, out_h):
    """im (H, W[, C]) float32 -> (out_h, out_w[, C]); cv2.INTER_CUBIC semantics (a = -0.75, replicated borders)."""
    im = np.asarray(im, dtype=np.float32)
    h, w = im.shape[:2]
    y0, fy = _axis_resample(h, out_h)
    x0, fx = _axis_resample(w, out_w)
    cy, cx = _cubic_coeffs(fy), _cubic_coeffs(fx)
    extra = (1,) * (im.ndim - 2)
    rows = np.zeros((out_h,) + im.shape[1:], dtype=np.float32)
    for k in range(4):
        rows += im[np.clip(y0 - 1 + k, 0, h - 1)] * cy[k].reshape((-1, 1) + extra)
    out = np.zeros((out_h, out_w) + im.shape[2:], dtype=np.float32)
    for k in range(4):
        out += rows[:, np.clip(x0 - 1 + k, 0, w - 1)] * cx[k].reshape((1, -1) + extra)
    return out


def move_batch_to_time(blob, num_frames):
    """(B*T, C, H, W) -> (B, C, T, H, W) (reference :82-93)."""
    bt, c, h, w = blob.shape
    assert bt % num_frames == 0
    return np.ascontiguousarray(blob.reshape(bt // num_frames, num_frames, c, h, w).transpose(0, 2, 1, 3, 4))
