"""Oracle: RPN proposal generation + FPN collect/distribute (test infrastructure).

Follows reference lib/ops/generate_proposals.py:40-196,
lib/ops/collect_and_distribute_fpn_rpn_proposals.py:44-87,
lib/modeling/FPN.py:349-381 and lib/ops/roi_blob_transforms.py:25-36.

Sort ties: reference uses unstable argpartition/argsort; the oracle defines
"higher score first, then lower flat (H, W, A) index" (see oracle/nms.py note).
"""
import numpy as np

from . import boxes as box_utils
from .anchors import all_shifted_anchors
from .nms import nms


def filter_boxes(boxes, min_size, im_info):
    """generate_proposals.py:184-196."""
    min_size = min_size * im_info[2]
    ws = boxes[:, 2] - boxes[:, 0] + 1
    hs = boxes[:, 3] - boxes[:, 1] + 1
    x_ctr = boxes[:, 0] + ws / 2.
    y_ctr = boxes[:, 1] + hs / 2.
    return np.where((ws >= min_size) & (hs >= min_size) &
                    (x_ctr < im_info[1]) & (y_ctr < im_info[0]))[0]


def proposals_for_one_image(im_info, all_anchors, bbox_deltas, scores, frames_per_vid,
                            pre_nms_topN, post_nms_topN, nms_thresh, min_size):
    """generate_proposals.py:40-114.  bbox_deltas (4*A*T, H, W), scores (A, H, W)."""
    bbox_deltas = bbox_deltas.transpose((1, 2, 0)).reshape((-1, 4 * frames_per_vid))
    scores = scores.transpose((1, 2, 0)).reshape((-1, 1))
    flat = scores.squeeze(1)
    order = np.argsort(-flat, kind='stable')
    if 0 < pre_nms_topN <= len(flat):
        order = order[:pre_nms_topN]
    bbox_deltas = bbox_deltas[order, :]
    all_anchors = all_anchors[order, :]
    scores = scores[order]
    proposals = box_utils.bbox_transform(all_anchors, bbox_deltas, (1.0, 1.0, 1.0, 1.0))
    proposals = box_utils.clip_tiled_boxes(proposals, im_info[:2])
    keep = np.arange(proposals.shape[0])
    for f in range(frames_per_vid):
        keep = np.intersect1d(keep, filter_boxes(proposals[:, f * 4:(f + 1) * 4], min_size, im_info))
    proposals = proposals[keep, :]
    scores = scores[keep]
    if nms_thresh > 0:
        keep = nms(np.hstack((proposals, scores)), nms_thresh)
        if post_nms_topN > 0:
            keep = keep[:post_nms_topN]
        proposals = proposals[keep, :]
        scores = scores[keep]
    return proposals, scores


def generate_proposals(scores, bbox_deltas, im_info, anchors, spatial_scale,
                       pre_nms_topN=1000, post_nms_topN=1000, nms_thresh=0.7, min_size=0):
    """GenerateProposalsOp.forward, generate_proposals.py:116-181.

    scores (N, A, H, W) fp32 probabilities, bbox_deltas (N, 4*A*T, H, W) fp32,
    im_info (N, 3).  Returns rois (R, 4T+1) fp32 and roi_probs (R, 1) fp32.
    """
    feat_stride = 1. / spatial_scale
    height, width = scores.shape[-2:]
    A = anchors.shape[0]
    assert bbox_deltas.shape[1] // A == anchors.shape[1]
    T = bbox_deltas.shape[1] // (4 * A)
    all_anchors = all_shifted_anchors(anchors, height, width, feat_stride)
    rois = np.empty((0, 4 * T + 1), dtype=np.float32)
    roi_probs = np.empty((0, 1), dtype=np.float32)
    for i in range(scores.shape[0]):
        b, p = proposals_for_one_image(im_info[i], all_anchors, bbox_deltas[i], scores[i], T,
                                       pre_nms_topN, post_nms_topN, nms_thresh, min_size)
        inds = i * np.ones((b.shape[0], 1), dtype=np.float32)
        rois = np.append(rois, np.hstack((inds, b)), axis=0)
        roi_probs = np.append(roi_probs, p, axis=0)
    return rois.astype(np.float32), roi_probs.astype(np.float32)


def map_rois_to_fpn_levels(rois, k_min, k_max, s0=224, lvl0=4):
    """FPN.py:349-360 (rois WITHOUT the batch column)."""
    s = np.sqrt(box_utils.boxes_area(rois))
    lvls = np.floor(lvl0 + np.log2(s / s0 + 1e-6))
    return np.clip(lvls, k_min, k_max)


def collect(roi_list, score_list, post_nms_topN):
    """collect_and_distribute_fpn_rpn_proposals.py:44-62."""
    rois = np.concatenate(roi_list)
    scores = np.concatenate(score_list).reshape(-1)
    inds = np.argsort(-scores, kind='stable')[:post_nms_topN]
    return rois[inds, :]


def distribute(rois, lvl_min, lvl_max):
    """collect_and_distribute_fpn_rpn_proposals.py:65-87.

    Returns (rois, [rois_fpn<lvl_min> ... rois_fpn<lvl_max>], idx_restore int32).
    """
    lvls = map_rois_to_fpn_levels(rois[:, 1:], lvl_min, lvl_max)
    per_level = []
    order = np.empty((0,))
    for lvl in range(lvl_min, lvl_max + 1):
        idx = np.where(lvls == lvl)[0]
        per_level.append(rois[idx, :])
        order = np.concatenate((order, idx))
    restore = np.argsort(order, kind='stable').astype(np.int32)
    return rois, per_level, restore


def roi_to_batch_format(rois):
    """(N, 4T+1) tube rois -> (N*T, 5) per-frame rois.  roi_blob_transforms.py:25-36."""
    T = (rois.shape[1] - 1) // 4
    N = rois.shape[0]
    out = np.zeros((N * T, 5))
    for t in range(T):
        out[t::T, 0] = rois[:, 0] * T + t
        out[t::T, 1:] = rois[:, 1 + 4 * t:1 + 4 * (t + 1)]
    return out
