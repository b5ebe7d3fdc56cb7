"""Oracle: RPN cell anchors (test infrastructure, see oracle/__init__.py).

Follows reference lib/modeling/generate_anchors.py:42-140.  All math is float64
as in the reference (`np.float`).
"""
import itertools

import numpy as np


def _whc(a):
    # generate_anchors.py:80-89
    w = a[2] - a[0] + 1
    h = a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _mk(ws, hs, cx, cy):
    # generate_anchors.py:92-105
    ws = np.asarray(ws, dtype=np.float64)[:, None]
    hs = np.asarray(hs, dtype=np.float64)[:, None]
    return np.hstack((cx - 0.5 * (ws - 1), cy - 0.5 * (hs - 1),
                      cx + 0.5 * (ws - 1), cy + 0.5 * (hs - 1)))


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512),
                     aspect_ratios=(0.5, 1, 2), time_dim=1,
                     tube_gen_style='replicate'):
    """(A, 4*time_dim) float64 anchors; generate_anchors.py:42-77."""
    scales = np.array(sizes, dtype=np.float64) / stride
    ratios = np.array(aspect_ratios, dtype=np.float64)
    base = np.array([1, 1, stride, stride], dtype=np.float64) - 1
    # ratio enumeration (generate_anchors.py:108-119)
    w, h, cx, cy = _whc(base)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    by_ratio = _mk(ws, hs, cx, cy)
    # scale enumeration per ratio anchor (generate_anchors.py:122-131)
    rows = []
    for a in by_ratio:
        w, h, cx, cy = _whc(a)
        rows.append(_mk(w * scales, h * scales, cx, cy))
    anchors = np.vstack(rows)
    # tube extension (generate_anchors.py:64-77)
    if tube_gen_style == 'replicate':
        anchors = np.tile(anchors, [1, time_dim])
    elif tube_gen_style == 'combinations':
        anchors = np.array([sum(c, []) for c in
                            itertools.combinations_with_replacement(anchors.tolist(), time_dim)])
    elif tube_gen_style == 'permutations':
        anchors = np.array([sum(c, []) for c in
                            itertools.permutations(anchors.tolist(), time_dim)])
    else:
        raise NotImplementedError(tube_gen_style)
    return anchors


def all_shifted_anchors(anchors, height, width, feat_stride):
    """(H*W*A, 4T) anchors at every cell, (H, W, A) slowest→fastest.

    ops/generate_proposals.py:139-165 (+ generate_anchors.py:134-140).
    """
    T = anchors.shape[1] // 4
    sx = np.arange(0, width) * feat_stride
    sy = np.arange(0, height) * feat_stride
    sx, sy = np.meshgrid(sx, sy)
    shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
    shifts = np.tile(shifts, [1, T])
    A = anchors.shape[0]
    K = shifts.shape[0]
    return (anchors[None, :, :] + shifts[:, None, :]).reshape(K * A, 4 * T)
