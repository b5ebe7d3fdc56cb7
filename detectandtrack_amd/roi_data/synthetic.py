"""Synthetic ground truth with the roidb record layout of lib/datasets/json_dataset.py:161-240 (there is no PoseTrack data
offline): `n_persons` boxes with 17 visible keypoints inside them, per clip, seeded."""
import numpy as np

from detectandtrack_amd.core.config import cfg


def synthetic_roidb_entry(height, width, n_persons=4, seed=0, T=1):
    rs = np.random.RandomState(seed)
    K = cfg.KRCNN.NUM_KEYPOINTS if cfg.KRCNN.NUM_KEYPOINTS > 0 else 17
    ncls = cfg.MODEL.NUM_CLASSES
    bw = rs.uniform(0.15, 0.45, n_persons) * width
    bh = rs.uniform(0.3, 0.8, n_persons) * height
    x1 = rs.uniform(0, 1, n_persons) * (width - bw - 1)
    y1 = rs.uniform(0, 1, n_persons) * (height - bh - 1)
    boxes = np.stack([x1, y1, x1 + bw, y1 + bh], axis=1).astype(np.float32)
    kps = np.zeros((n_persons, 3, K), dtype=np.int32)
    kps[:, 0, :] = (x1[:, None] + rs.uniform(0.05, 0.95, (n_persons, K)) * bw[:, None]).astype(np.int32)
    kps[:, 1, :] = (y1[:, None] + rs.uniform(0.05, 0.95, (n_persons, K)) * bh[:, None]).astype(np.int32)
    kps[:, 2, :] = 2
    ov = np.zeros((n_persons, ncls), dtype=np.float32)
    ov[:, 1] = 1.0
    if T > 1:   # tubes: the person drifts a few pixels per frame; keypoints move with the box
        bx, kp = [boxes], [kps]
        for t in range(1, T):
            d = rs.uniform(-4, 4, (n_persons, 2)).astype(np.float32)
            nb = np.clip(bx[-1] + np.tile(d, (1, 2)), 0, [width - 1, height - 1, width - 1, height - 1]).astype(np.float32)
            nk = kp[-1].copy()
            nk[:, 0, :] = np.clip(nk[:, 0, :] + d[:, 0:1].astype(np.int32), 0, width - 1)
            nk[:, 1, :] = np.clip(nk[:, 1, :] + d[:, 1:2].astype(np.int32), 0, height - 1)
            bx.append(nb)
            kp.append(nk)
        boxes, kps = np.concatenate(bx, axis=1), np.concatenate(kp, axis=2)
    return dict(height=height, width=width, boxes=boxes, gt_classes=np.ones((n_persons,), np.int32),
                is_crowd=np.zeros((n_persons,), np.bool_), gt_overlaps=ov,
                box_to_gt_ind_map=np.arange(n_persons, dtype=np.int32), gt_keypoints=kps,
                max_overlaps=ov.max(axis=1), max_classes=ov.argmax(axis=1))
