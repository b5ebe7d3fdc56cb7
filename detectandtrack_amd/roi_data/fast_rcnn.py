"""Fast R-CNN / keypoint training blobs sampled from the RPN proposals of the current clip (reference
lib/roi_data/fast_rcnn.py:109-229, lib/roi_data/keypoint_rcnn.py:32-99, lib/datasets/json_dataset.py:423-535).

`sample_training_blobs` is what the training executor calls from CollectAndDistributeFpnRpnProposals
(lib/ops/collect_and_distribute_fpn_rpn_proposals.py:24-41): merge the proposals into the clip's roidb entry, sample
BATCH_SIZE_PER_IM rois (FG_FRACTION foreground at IoU >= FG_THRESH), expand the class-specific box targets, pick the
keypoint rois and turn their ground-truth keypoints into heatmap cell indices.
"""
import numpy as np
import numpy.random as npr

from detectandtrack_amd.core.config import cfg
import detectandtrack_amd.utils.boxes as box_utils


def merge_proposals_into_entry(entry, proposals):
    """json_dataset.py:423-473 + :_add_class_assignments: returns a NEW dict with the proposals appended to the gt boxes."""
    e = dict(entry)
    n = proposals.shape[0]
    ncls = entry['gt_overlaps'].shape[1]
    gt_inds = np.where(entry['gt_classes'] > 0)[0]
    ov = np.zeros((n, ncls), dtype=np.float32)
    b2g = -np.ones((n,), dtype=np.int32)
    if len(gt_inds) > 0 and n > 0:
        p2g = box_utils.bbox_overlaps(proposals.astype(np.float32), entry['boxes'][gt_inds].astype(np.float32))
        arg, mx = p2g.argmax(axis=1), p2g.max(axis=1)
        nz = np.where(mx > 0)[0]
        ov[nz, entry['gt_classes'][gt_inds][arg[nz]]] = mx[nz]
        b2g[nz] = gt_inds[arg[nz]]
    e['boxes'] = np.append(entry['boxes'], proposals.astype(entry['boxes'].dtype), axis=0)
    e['gt_classes'] = np.append(entry['gt_classes'], np.zeros((n,), dtype=entry['gt_classes'].dtype))
    e['is_crowd'] = np.append(entry['is_crowd'], np.zeros((n,), dtype=entry['is_crowd'].dtype))
    e['gt_overlaps'] = np.append(np.asarray(entry['gt_overlaps']), ov, axis=0)
    e['box_to_gt_ind_map'] = np.append(entry['box_to_gt_ind_map'], b2g)
    e['max_overlaps'] = e['gt_overlaps'].max(axis=1)
    e['max_classes'] = e['gt_overlaps'].argmax(axis=1)
    return e


def expand_bbox_targets(target_data):
    """:206-229: (label, 4T targets) rows -> 4T-of-(4T*K) class-specific targets + inside weights."""
    tube = target_data.shape[-1] - 1
    K = 2 if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG else cfg.MODEL.NUM_CLASSES
    cls = target_data[:, 0]
    targets = np.zeros((cls.size, tube * K), dtype=np.float32)
    w_in = np.zeros(targets.shape, dtype=np.float32)
    for i in np.where(cls > 0)[0]:
        s = tube * int(cls[i])
        targets[i, s:s + tube] = target_data[i, 1:]
        w_in[i, s:s + tube] = 1.0
    return targets, w_in


def keypoints_to_heatmap_labels(keypoints, rois):
    """utils/keypoints.py:152-207: (N, 3, K) keypoints + (N, 4) rois -> heatmap cell index and weight per keypoint."""
    M = cfg.KRCNN.HEATMAP_SIZE
    heat = np.zeros((len(rois), cfg.KRCNN.NUM_KEYPOINTS), dtype=np.float32)
    wts = np.zeros_like(heat)
    sx = M / (rois[:, 2] - rois[:, 0] + 1)
    sy = M / (rois[:, 3] - rois[:, 1] + 1)
    for kp in range(keypoints.shape[2]):
        vis = keypoints[:, 2, kp] > 0
        x = keypoints[:, 0, kp].astype(np.float32)
        y = keypoints[:, 1, kp].astype(np.float32)
        xb, yb = np.where(x == rois[:, 2])[0], np.where(y == rois[:, 3])[0]
        x = np.floor((x - rois[:, 0]) * sx)
        y = np.floor((y - rois[:, 1]) * sy)
        x[xb] = M - 1
        y[yb] = M - 1
        valid = ((x >= 0) & (y >= 0) & (x < M) & (y < M) & vis).astype(np.int32)
        heat[:, kp] = (y * M + x) * valid
        wts[:, kp] = valid
    return heat, wts


def _within_box(points, boxes):
    return ((points[:, 0, :] >= boxes[:, 0:1]) & (points[:, 0, :] <= boxes[:, 2:3]) &
            (points[:, 1, :] >= boxes[:, 1:2]) & (points[:, 1, :] <= boxes[:, 3:4]))


def add_keypoint_blobs(blobs, e, fg_rois_per_image, im_scale, batch_idx, rng):
    """roi_data/keypoint_rcnn.py:32-86.  Tubes: boxes are (n, 4T), keypoints (n, 3, K*T); the visibility / within-box test
    uses the FIRST frame's box against all T*K keypoints exactly like the reference's _within_box (:88-99), the heatmap
    targets are built per frame and concatenated along the keypoint axis (:62-73)."""
    gt_inds = np.where(e['gt_classes'] > 0)[0]
    kp_fg = keypoint_fg_candidates(e)
    n = min(fg_rois_per_image, kp_fg.size)
    if kp_fg.size > n:
        kp_fg = rng.choice(kp_fg, size=n, replace=False)
    if kp_fg.shape[0] == 0:
        kp_fg = gt_inds
    keypoint_blobs_for(blobs, e, kp_fg, im_scale, batch_idx)


def keypoint_fg_candidates(e):
    """roi_data/keypoint_rcnn.py:40-46: foreground rois that see a visible keypoint of their ground-truth box (the set the draw is from)."""
    gt_inds = np.where(e['gt_classes'] > 0)[0]
    gtk = e['gt_keypoints']
    ind_kp = gt_inds[e['box_to_gt_ind_map']]
    within = _within_box(gtk[ind_kp], e['boxes'])
    visible = np.sum((gtk[ind_kp, 2, :] > 0) & within, axis=1) > 0
    return np.where((e['max_overlaps'] >= cfg.TRAIN.FG_THRESH) & visible)[0]


def keypoint_blobs_for(blobs, e, kp_fg, im_scale, batch_idx):
    """The keypoint blobs of the DRAWN keypoint rois `kp_fg` (indices into the merged entry): roi_data/keypoint_rcnn.py:49-86."""
    gtk = e['gt_keypoints']
    rois = e['boxes'][kp_fg].astype(np.float32)
    b2g = e['box_to_gt_ind_map'][kp_fg]
    kps = -np.ones((len(rois), gtk.shape[1], gtk.shape[2]), dtype=gtk.dtype)
    for i in range(len(rois)):
        if b2g[i] >= 0:
            kps[i] = gtk[b2g[i]]
    T = rois.shape[-1] // 4
    K = gtk.shape[2] // T
    heats, wts = [], []
    for t in range(T):
        h, w = keypoints_to_heatmap_labels(kps[..., t * K:(t + 1) * K], rois[..., t * 4:(t + 1) * 4])
        heats.append(h)
        wts.append(w)
    heat, wt = np.concatenate(heats, axis=-1), np.concatenate(wts, axis=-1)
    blobs['keypoint_rois'] = np.hstack((batch_idx * np.ones((len(rois), 1), np.float32), rois * im_scale)).astype(np.float32)
    blobs['keypoint_locations_int32'] = heat.reshape(-1, 1).astype(np.int32)
    blobs['keypoint_weights'] = wt.reshape(-1, 1).astype(np.float32)
    blobs['keypoint_loss_normalizer'] = np.array([1.0], dtype=np.float32)


def sample_rois(e, im_scale, batch_idx, rng):
    """:129-203"""
    per_im = int(cfg.TRAIN.BATCH_SIZE_PER_IM)
    fg_per_im = int(np.round(cfg.TRAIN.FG_FRACTION * per_im))
    mo = e['max_overlaps']
    fg = np.where(mo >= cfg.TRAIN.FG_THRESH)[0]
    n_fg = min(fg_per_im, fg.size)
    if fg.size > 0:
        fg = rng.choice(fg, size=n_fg, replace=False)
    bg = np.where((mo < cfg.TRAIN.BG_THRESH_HI) & (mo >= cfg.TRAIN.BG_THRESH_LO))[0]
    n_bg = min(per_im - n_fg, bg.size)
    if bg.size > 0:
        bg = rng.choice(bg, size=n_bg, replace=False)
    keep = np.append(fg, bg).astype(np.int64)
    blobs = roi_blobs_for(e, keep, n_fg, im_scale, batch_idx)
    if cfg.MODEL.KEYPOINTS_ON:
        add_keypoint_blobs(blobs, e, fg_per_im, im_scale, batch_idx, rng)
    return blobs


def roi_blobs_for(e, keep, n_fg, im_scale, batch_idx):
    """The Fast R-CNN blobs of the DRAWN rois `keep` (indices into the merged entry, the first n_fg of them foreground): :156-203."""
    labels = e['max_classes'][keep].copy()
    labels[n_fg:] = 0
    boxes = e['boxes'][keep].astype(np.float32)
    gt_inds = np.where(e['gt_classes'] > 0)[0]
    gt_boxes = e['boxes'][gt_inds].astype(np.float32)
    assign = e['box_to_gt_ind_map'][keep]
    tgt = np.zeros((len(keep), boxes.shape[1]), dtype=np.float32)
    has = assign >= 0
    if has.any():
        tgt[has] = box_utils.bbox_transform_inv(boxes[has], e['boxes'][assign[has]].astype(np.float32),
                                                cfg.MODEL.BBOX_REG_WEIGHTS)
    targets, w_in = expand_bbox_targets(np.hstack((labels[:, None].astype(np.float32), tgt)))
    blobs = dict(labels_int32=labels.astype(np.int32),
                 rois=np.hstack((batch_idx * np.ones((len(keep), 1), np.float32), boxes * im_scale)).astype(np.float32),
                 bbox_targets=targets, bbox_inside_weights=w_in, bbox_outside_weights=(w_in > 0).astype(np.float32))
    del gt_boxes
    return blobs


def sample_training_blobs(entry, rois, im_info, rng=npr):
    """rois: (R, 5) proposals at network scale (batch idx 0); entry: the clip's roidb record at original scale."""
    scale = float(im_info[0, 2])
    e = merge_proposals_into_entry(entry, rois[:, 1:] / scale)
    return sample_rois(e, scale, 0, rng)
