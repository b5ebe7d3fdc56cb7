"""Keypoint utilities — mirror of reference lib/utils/keypoints.py (heatmap decoding :94-149, :210-216)."""
import numpy as np

from detectandtrack_amd.core.config import cfg
import detectandtrack_amd.utils.image as image_utils


def get_keypoints():
    """COCO person keypoints and the left/right flip map (reference :23-60)."""
    names = ['nose', 'left_eye', 'right_eye', 'left_ear', 'right_ear', 'left_shoulder', 'right_shoulder',
             'left_elbow', 'right_elbow', 'left_wrist', 'right_wrist', 'left_hip', 'right_hip', 'left_knee',
             'right_knee', 'left_ankle', 'right_ankle']
    flip = {n: n.replace('left', 'right') for n in names if n.startswith('left')}
    return names, flip


def get_person_class_index():
    return 1


def scores_to_probs(scores):
    """Spatial softmax per channel of a CxHxW score map (:210-216)."""
    for c in range(scores.shape[0]):
        t = scores[c]
        e = np.exp(t - t.max())
        scores[c] = e / np.sum(e)
    return scores


def heatmaps_to_keypoints(maps, rois):
    """(#rois, K, M, M) heatmaps + (#rois, 4) boxes -> (#rois, 4, K) rows (x, y, logit, prob) (:94-149): each map
    is bicubically resized to the RoI's (ceil) size, the argmax cell centre is mapped back to image coordinates."""
    off_x, off_y = rois[:, 0], rois[:, 1]
    widths = np.maximum(rois[:, 2] - rois[:, 0], 1)
    heights = np.maximum(rois[:, 3] - rois[:, 1], 1)
    wc, hc = np.ceil(widths), np.ceil(heights)
    maps = np.transpose(maps, [0, 2, 3, 1])
    min_size = cfg.KRCNN.INFERENCE_MIN_SIZE
    K = cfg.KRCNN.NUM_KEYPOINTS
    xy = np.zeros((len(rois), 4, K), dtype=np.float32)
    for i in range(len(rois)):
        mw = int(np.maximum(wc[i], min_size)) if min_size > 0 else int(wc[i])
        mh = int(np.maximum(hc[i], min_size)) if min_size > 0 else int(hc[i])
        w_corr, h_corr = widths[i] / mw, heights[i] / mh
        roi_map = np.transpose(image_utils.resize_bicubic(maps[i], mw, mh), [2, 0, 1])
        probs = scores_to_probs(roi_map.copy())
        w = roi_map.shape[2]
        for k in range(K):
            pos = roi_map[k].argmax()
            x_int = pos % w
            y_int = (pos - x_int) // w
            xy[i, 0, k] = (x_int + 0.5) * w_corr + off_x[i]
            xy[i, 1, k] = (y_int + 0.5) * h_corr + off_y[i]
            xy[i, 2, k] = roi_map[k, y_int, x_int]
            xy[i, 3, k] = probs[k, y_int, x_int]
    return xy
