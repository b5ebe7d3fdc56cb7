"""Oracle: the host glue between `model.net` and `model.keypoint_net` (test infrastructure, see oracle/__init__.py).

Restates, in NumPy float32 with the reference's operation order,
  lib/core/test.py:211-244   im_detect_bbox tail: `boxes = rois[:, 1:] / im_scales[0]`, bbox_transform with
                             MODEL.BBOX_REG_WEIGHTS (lib/utils/boxes.py:141-202 -> oracle/boxes.py), clip_tiled_boxes;
  lib/core/test.py:750-806   box_results_with_nms_and_limit (score threshold, per-class NMS through lib/core/nms_wrapper.py,
                             DETECTIONS_PER_IM over all classes with `>=` the D-th best score);
  lib/core/test.py:78-123    _get_rois_blob / _project_im_rois (single scale: level 0, boxes * scale in float64, cast to float32).

NumPy-version note: the reference environment pins numpy 1.14.2 (all_pkg_versions.txt:164), whose value-based casting makes
`float32_array / np.float64_scalar` a FLOAT32 division by float32(scale); NumPy 2 would promote to float64.  The oracle follows
the reference environment.  PARITY: bbox_transform / clip / NMS are pinned (golden vectors from the real reference,
tests/test_oracle_golden.py; the Cython NMS in oracle/_ref); the glue around them is restated from the cited lines.
"""
import numpy as np

from . import boxes as obox
from . import nms as onms


def read_bbox_outputs(rois, cls_prob, bbox_pred, im_scale, im_shape, reg_weights=(10., 10., 5., 5.), cls_agnostic=False):
    """test.py:211-244.  rois (R, 4T+1) network coordinates, im_shape = (H, W) of the unscaled image.
    Returns scores (R, K), pred_boxes (R, K*4T)."""
    rois = np.asarray(rois, dtype=np.float32)
    boxes = rois[:, 1:] / np.float32(im_scale)                       # float32 / float32 (numpy 1.14 value-based casting)
    scores = np.asarray(cls_prob, dtype=np.float32).reshape([-1, cls_prob.shape[-1]])
    time_dim = boxes.shape[-1] // 4
    box_deltas = np.asarray(bbox_pred, dtype=np.float32).reshape([-1, bbox_pred.shape[-1]])
    if cls_agnostic:
        box_deltas = box_deltas[:, -4 * time_dim:]
    pred_boxes = obox.bbox_transform(boxes, box_deltas, reg_weights)
    pred_boxes = obox.clip_tiled_boxes(pred_boxes, im_shape)
    if cls_agnostic:
        pred_boxes = np.tile(pred_boxes, (1, scores.shape[1]))
    return scores, pred_boxes


def box_results_with_nms_and_limit(scores, boxes, num_classes, score_thresh=0.05, nms_thresh=0.5, detections_per_im=100):
    """test.py:750-806 (Soft-NMS and box voting disabled, as in every shipped config)."""
    time_dim = boxes.shape[-1] // (num_classes * 4)
    cls_boxes = [[] for _ in range(num_classes)]
    for j in range(1, num_classes):
        inds = np.where(scores[:, j] > score_thresh)[0]
        scores_j = scores[inds, j]
        boxes_j = boxes[inds, j * 4 * time_dim:(j + 1) * 4 * time_dim]
        dets_j = np.hstack((boxes_j, scores_j[:, np.newaxis])).astype(np.float32, copy=False)
        keep = onms.nms(dets_j, nms_thresh)
        cls_boxes[j] = dets_j[np.asarray(keep, dtype=np.int64), :]
    if detections_per_im > 0:
        image_scores = np.hstack([cls_boxes[j][:, -1] for j in range(1, num_classes)])
        if len(image_scores) > detections_per_im:
            image_thresh = np.sort(image_scores)[-detections_per_im]
            for j in range(1, num_classes):
                keep = np.where(cls_boxes[j][:, -1] >= image_thresh)[0]
                cls_boxes[j] = cls_boxes[j][keep, :]
    im_results = np.vstack([cls_boxes[j] for j in range(1, num_classes)])
    return im_results[:, -1], im_results[:, :-1], cls_boxes


def get_rois_blob(im_rois, im_scale):
    """test.py:78-123, one scale: [level 0, boxes * scale] with the product formed in float64."""
    rois = im_rois.astype(np.float64, copy=False) * np.float64(im_scale)
    return np.hstack((np.zeros((im_rois.shape[0], 1)), rois)).astype(np.float32, copy=False)
