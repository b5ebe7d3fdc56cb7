"""Oracle: legacy (Detectron/Caffe2) RoIAlign, 2D and tube (test infrastructure).

The operator itself lives in Caffe2 modules/detectron (commit b4e1588, not in
/root/reference) — PARITY UNPINNED.  Restated from its public semantics as used
at the reference call sites lib/modeling/detector.py:216-254 (tube handling:
RoIToBatchFormat + time->batch, detector.py:228-233, ops/roi_blob_transforms.py)
and detector.py:240-245 (pooled_w/h, spatial_scale, sampling_ratio):

  * roi corners scaled by spatial_scale, NO half-pixel shift, no rounding;
  * roi_w = max(x2-x1, 1), roi_h = max(y2-y1, 1); bin = roi / pooled;
  * grid = sampling_ratio if > 0 else ceil(roi / pooled); samples at
    start + p*bin + (i + .5)*bin/grid, averaged;
  * a sample with y < -1 or y > H or x < -1 or x > W contributes 0; otherwise
    coordinates are clamped to [0, size-1] before bilinear interpolation.
"""
import numpy as np


def _bilinear(feat, y, x):
    """feat (C, H, W) float32; scalar y, x. Returns (C,) float32."""
    C, H, W = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return np.zeros((C,), dtype=np.float32)
    y = np.float32(max(y, 0.0))
    x = np.float32(max(x, 0.0))
    y_low = int(y)
    x_low = int(x)
    if y_low >= H - 1:
        y_high = y_low = H - 1
        y = np.float32(y_low)
    else:
        y_high = y_low + 1
    if x_low >= W - 1:
        x_high = x_low = W - 1
        x = np.float32(x_low)
    else:
        x_high = x_low + 1
    ly = np.float32(y - np.float32(y_low))
    lx = np.float32(x - np.float32(x_low))
    hy = np.float32(1.) - ly
    hx = np.float32(1.) - lx
    return (hy * hx * feat[:, y_low, x_low] + hy * lx * feat[:, y_low, x_high] +
            ly * hx * feat[:, y_high, x_low] + ly * lx * feat[:, y_high, x_high]).astype(np.float32)


def roi_align_2d(feat, rois, pooled, spatial_scale, sampling_ratio):
    """feat (N, C, H, W) fp32, rois (R, 5) [batch, x1, y1, x2, y2] -> (R, C, P, P)."""
    feat = np.asarray(feat, dtype=np.float32)
    R = rois.shape[0]
    C = feat.shape[1]
    out = np.zeros((R, C, pooled, pooled), dtype=np.float32)
    sc = np.float32(spatial_scale)
    for r in range(R):
        b = int(rois[r, 0])
        x1 = np.float32(rois[r, 1]) * sc
        y1 = np.float32(rois[r, 2]) * sc
        x2 = np.float32(rois[r, 3]) * sc
        y2 = np.float32(rois[r, 4]) * sc
        rw = np.float32(max(x2 - x1, np.float32(1.)))
        rh = np.float32(max(y2 - y1, np.float32(1.)))
        bh = rh / np.float32(pooled)
        bw = rw / np.float32(pooled)
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rh / pooled))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rw / pooled))
        cnt = np.float32(gh * gw)
        for ph in range(pooled):
            for pw in range(pooled):
                acc = np.zeros((C,), dtype=np.float32)
                for iy in range(gh):
                    y = y1 + np.float32(ph) * bh + np.float32(iy + .5) * bh / np.float32(gh)
                    for ix in range(gw):
                        x = x1 + np.float32(pw) * bw + np.float32(ix + .5) * bw / np.float32(gw)
                        acc += _bilinear(feat[b], float(y), float(x))
                out[r, :, ph, pw] = acc / cnt
    return out


def roi_align_tube(feat5, rois, pooled, spatial_scale, sampling_ratio):
    """feat5 (N, C, T, H, W); rois (R, 4T+1) tube rois -> (R, C, T, P, P).

    detector.py:216-254: rois -> (R*T, 5) with batch n*T+t; features time->batch;
    2D RoIAlign; reshape back to R x C x T x P x P.
    """
    from .proposals import roi_to_batch_format
    N, C, T, H, W = feat5.shape
    rois_b = roi_to_batch_format(rois)
    feat_b = np.ascontiguousarray(feat5.transpose(0, 2, 1, 3, 4)).reshape(N * T, C, H, W)
    out = roi_align_2d(feat_b, rois_b, pooled, spatial_scale, sampling_ratio)
    R = rois.shape[0]
    return np.ascontiguousarray(out.reshape(R, T, C, pooled, pooled).transpose(0, 2, 1, 3, 4))
