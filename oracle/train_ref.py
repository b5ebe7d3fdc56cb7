"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): torch-autograd restatement of the reference's TRAINING graph for the
FPN / 2D-head keypoint R-CNN, used to check the HIP backward pass parameter by parameter.

Follows lib/modeling/model_builder.py:179-306 (train branch), losses :481-494, :873-889, FPN.py:282-321; loss op
semantics restated from the public Caffe2 / Detectron operators (SigmoidCrossEntropyLoss with -1 = ignore,
SmoothL1Loss with inside/outside weights normalised by dim 0, SoftmaxWithLoss normalised by N or by the weight sum).
Parity status: unpinned (no reference implementation of these Caffe2 ops exists in the tree).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import proposals as prop
from .net3d import Net


def roi_align_2d_torch(feat, rois, pooled, spatial_scale, sampling_ratio):
    """Differentiable legacy RoIAlign (oracle/roi_align.py roi_align_2d restated with torch gathers).
    feat (N, C, H, W) tensor, rois (R, 5) ndarray -> (R, C, P, P)."""
    N, C, H, W = feat.shape
    outs = []
    sc = np.float32(spatial_scale)
    for r in range(rois.shape[0]):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = [np.float32(v) * sc for v in rois[r, 1:5]]
        rw = np.float32(max(x2 - x1, np.float32(1.)))
        rh = np.float32(max(y2 - y1, np.float32(1.)))
        bh, bw = rh / np.float32(pooled), rw / np.float32(pooled)
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rh / pooled))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rw / pooled))
        ys = (y1 + np.arange(pooled, dtype=np.float32)[:, None] * bh +
              (np.arange(gh, dtype=np.float32)[None, :] + np.float32(.5)) * bh / np.float32(gh)).reshape(-1)
        xs = (x1 + np.arange(pooled, dtype=np.float32)[:, None] * bw +
              (np.arange(gw, dtype=np.float32)[None, :] + np.float32(.5)) * bw / np.float32(gw)).reshape(-1)

        def axis(v, size):
            valid = (v >= -1.0) & (v <= size)
            v = np.maximum(v, 0)
            lo = v.astype(np.int64)
            hi = lo + 1
            edge = lo >= size - 1
            lo = np.where(edge, size - 1, lo)
            hi = np.where(edge, size - 1, hi)
            v = np.where(edge, lo.astype(np.float32), v)
            l = (v - lo).astype(np.float32)
            return lo, hi, l, valid
        ylo, yhi, ly, yv = axis(ys, H)
        xlo, xhi, lx, xv = axis(xs, W)
        f = feat[b]
        ly_t, lx_t = torch.from_numpy(ly)[:, None], torch.from_numpy(lx)[None, :]
        v = ((1 - ly_t) * (1 - lx_t)) * f[:, ylo][:, :, xlo] + ((1 - ly_t) * lx_t) * f[:, ylo][:, :, xhi] + \
            (ly_t * (1 - lx_t)) * f[:, yhi][:, :, xlo] + (ly_t * lx_t) * f[:, yhi][:, :, xhi]
        v = v * torch.from_numpy((yv[:, None] & xv[None, :]).astype(np.float32))
        v = v.view(C, pooled, gh, pooled, gw).sum(dim=(2, 4)) / float(gh * gw)
        outs.append(v)
    return torch.stack(outs) if outs else feat.new_zeros((0, C, pooled, pooled))


def roi_feat_fpn_torch(levels_p5_to_p2, rois, pooled, sampling):
    lvls = prop.map_rois_to_fpn_levels(rois[:, 1:], 2, 5)
    out = [None] * rois.shape[0]
    for lvl in range(2, 6):
        idx = np.where(lvls == lvl)[0]
        if len(idx) == 0:
            continue
        f = roi_align_2d_torch(levels_p5_to_p2[5 - lvl], rois[idx], pooled, 1. / 2. ** lvl, sampling)
        for j, i in enumerate(idx):
            out[i] = f[j]
    return torch.stack(out)


def smooth_l1(pred, tgt, w_in, w_out, beta):
    v = w_in * (pred - tgt)
    return (w_out * torch.where(v.abs() < beta, 0.5 * v * v / beta, v.abs() - 0.5 * beta)).sum()


def training_losses(weights, opts, data, im_info, labels, sampled, cfg_scalars):
    """weights: dict name -> torch leaf tensors; labels: dict of the 'wide' RPN label arrays per level; sampled: dict with
    rois, labels_int32, bbox_targets, bbox_inside_weights, bbox_outside_weights, keypoint_rois, keypoint_locations_int32,
    keypoint_weights.  cfg_scalars: num_gpus, rpn_batch, ims_per_batch, kps_loss_weight.  Returns dict loss name -> tensor."""
    net = Net(weights, opts)
    net.body(torch.from_numpy(data))
    p2d = net.time_link(net.fpn())                    # [P6, P5, P4, P3, P2] 2-D maps
    ng = cfg_scalars['num_gpus']
    losses = {}
    for lvl in range(2, 7):
        x = p2d[6 - lvl]
        h = F.relu(net.conv2d(x, 'conv_rpn_fpn2', 3, 1, 1))
        logits = net.conv2d(h, 'rpn_cls_logits_fpn2', 1)
        deltas = net.conv2d(h, 'rpn_bbox_pred_fpn2', 1)
        H, W = logits.shape[2:]
        lab = torch.from_numpy(labels['rpn_labels_int32_wide_fpn%d' % lvl][:, :, :H, :W])
        valid = (lab >= 0).float()
        ce = F.binary_cross_entropy_with_logits(logits, lab.clamp(min=0).float(), reduction='none')
        losses['loss_rpn_cls_fpn%d' % lvl] = (ce * valid).sum() / ng / cfg_scalars['rpn_batch'] / cfg_scalars['ims_per_batch']
        t, wi, wo = [torch.from_numpy(labels['rpn_bbox_%s_wide_fpn%d' % (k, lvl)][:, :, :H, :W])
                     for k in ('targets', 'inside_weights', 'outside_weights')]
        losses['loss_rpn_bbox_fpn%d' % lvl] = smooth_l1(deltas, t, wi, wo, 1. / 9.) / logits.shape[0] / ng
    rois = sampled['rois']
    feat = roi_feat_fpn_torch(p2d[1:], rois, opts['frcn_res'], opts['frcn_sampling'])
    x = F.relu(net.fc(feat, 'fc6'))
    x = F.relu(net.fc(x, 'fc7'))
    cls_score, bbox_pred = net.fc(x, 'cls_score'), net.fc(x, 'bbox_pred')
    R = rois.shape[0]
    losses['loss_cls'] = F.cross_entropy(cls_score, torch.from_numpy(sampled['labels_int32']).long(), reduction='sum') / R / ng
    losses['loss_bbox'] = smooth_l1(bbox_pred, torch.from_numpy(sampled['bbox_targets']),
                                    torch.from_numpy(sampled['bbox_inside_weights']),
                                    torch.from_numpy(sampled['bbox_outside_weights']), 1.0) / R / ng
    kfeat = roi_feat_fpn_torch(p2d[1:], sampled['keypoint_rois'], opts['kps_res'], opts['kps_sampling'])
    kps = net.kps_head_2d(kfeat)                      # (Rk, K, M, M)
    Rk, K, M, _ = kps.shape
    w = torch.from_numpy(sampled['keypoint_weights']).reshape(-1)
    nll = F.cross_entropy(kps.reshape(Rk * K, M * M), torch.from_numpy(sampled['keypoint_locations_int32']).reshape(-1).long(),
                          reduction='none')
    losses['loss_kps'] = (nll * w).sum() / w.sum() * cfg_scalars['kps_loss_weight'] / ng
    return losses


def roi_align_tube_torch(feat5, rois, pooled, spatial_scale, sampling):
    """feat5 (N, C, T, H, W) tensor; rois (R, 4T+1) -> (R, C, T, P, P): frame t of the tube pools frame t of the features."""
    T = feat5.shape[2]
    outs = []
    for t in range(T):
        r = np.hstack((rois[:, :1], rois[:, 1 + 4 * t:5 + 4 * t])).astype(np.float32)
        outs.append(roi_align_2d_torch(feat5[:, :, t], r, pooled, spatial_scale, sampling))
    return torch.stack(outs, dim=2)


def training_losses_c4_tube(weights, opts, data, labels, sampled, cfg_scalars):
    """The shipped 3D configuration in training mode (model_builder.py:179-306 train branch with ResNet3D C4 body, tube RPN
    :500-636, per-RoI res5 head ResNet3D.py:301-327 + :426-494, 3D keypoint head :755-889)."""
    net = Net(weights, opts)
    feat = net.body(torch.from_numpy(data))           # (1, C, T, H, W)
    o = opts
    T, kt = o['num_frames_mid'], o['kt_rpn']
    ng = cfg_scalars['num_gpus']
    losses = {}
    h = F.relu(net.conv_nd(feat, 'conv_rpn', [kt, 3, 3], [1, 1, 1], [kt // 2, 1, 1]))
    logits = net.conv_nd(h, 'rpn_cls_logits_1', [1, 1, 1], [1, 1, 1], [0, 0, 0]).mean(dim=2)
    d = net.conv_nd(h, 'rpn_bbox_pred_1', [1, 1, 1], [1, 1, 1], [0, 0, 0])
    N, A4, Tt, H, W = d.shape
    A = A4 // 4
    d = d.reshape(N, A, 4, Tt, H, W).permute(0, 1, 3, 2, 4, 5).reshape(N, A * Tt * 4, H, W)
    lab = torch.from_numpy(labels['rpn_labels_int32_wide'][:, :, :H, :W])
    valid = (lab >= 0).float()
    ce = F.binary_cross_entropy_with_logits(logits, lab.clamp(min=0).float(), reduction='none')
    losses['loss_rpn_cls'] = (ce * valid).sum() / max(float(valid.sum()), 1.0) / ng
    t, wi, wo = [torch.from_numpy(labels['rpn_bbox_%s_wide' % k][:, :, :H, :W])
                 for k in ('targets', 'inside_weights', 'outside_weights')]
    losses['loss_rpn_bbox'] = smooth_l1(d, t, wi, wo, 1. / 9.) / N / ng / T
    rois = sampled['rois']
    R = rois.shape[0]
    pooled = o['frcn_res']
    x = roi_align_tube_torch(feat, rois, pooled, 1. / 16., o['frcn_sampling'])
    dims = o['feat_dims']
    x = net._stage(x, 4, 'res5', o['res5_blocks'], dims[3], o['res5_dim'], 1, stride_init=int(pooled / 7))
    x = x.mean(dim=4).mean(dim=3)[:, :, :, None, None]
    cls = net.conv_nd(x, 'cls_score_1', [1, 1, 1], [1, 1, 1], [0, 0, 0]).mean(dim=4).mean(dim=3).mean(dim=2)
    bp = net.conv_nd(x, 'bbox_pred_1', [1, 1, 1], [1, 1, 1], [0, 0, 0])
    K4 = bp.shape[1]
    bp = bp.reshape(R, K4 // 4, 4, T, 1, 1).permute(0, 1, 3, 2, 4, 5).reshape(R, -1)
    losses['loss_cls'] = F.cross_entropy(cls, torch.from_numpy(sampled['labels_int32']).long(), reduction='sum') / R / ng
    losses['loss_bbox'] = smooth_l1(bp, torch.from_numpy(sampled['bbox_targets']), torch.from_numpy(sampled['bbox_inside_weights']),
                                    torch.from_numpy(sampled['bbox_outside_weights']), 1.0) / R / ng / T
    kx = roi_align_tube_torch(feat, sampled['keypoint_rois'], o['kps_res'], 1. / 16., o['kps_sampling'])
    kps = net.kps_head_tube_feat(kx)                  # (Rk, T*K, M, M)
    Rk, TK, M, _ = kps.shape
    w = torch.from_numpy(sampled['keypoint_weights']).reshape(-1)
    nll = F.cross_entropy(kps.reshape(Rk * TK, M * M), torch.from_numpy(sampled['keypoint_locations_int32']).reshape(-1).long(),
                          reduction='none')
    losses['loss_kps'] = (nll * w).sum() / w.sum() * cfg_scalars['kps_loss_weight'] / ng
    return losses


def roi_feat_fpn_tube_torch(pyr_p5_to_p2, rois, pooled, sampling):
    """Multi-level tube RoIAlign with autograd: level by the tube's mean area (FPN.py:349-360), frame t of the tube pools
    frame t of that level (detector.py:256-310) -> (R, C, T, P, P)."""
    lvls = prop.map_rois_to_fpn_levels(rois[:, 1:], 2, 5)
    out = [None] * rois.shape[0]
    for lvl in range(2, 6):
        idx = np.where(lvls == lvl)[0]
        if len(idx) == 0:
            continue
        f = roi_align_tube_torch(pyr_p5_to_p2[5 - lvl], rois[idx], pooled, 1. / 2. ** lvl, sampling)
        for j, i in enumerate(idx):
            out[i] = f[j]
    return torch.stack(out)


def training_losses_fpn_tube(weights, opts, data, labels, sampled, cfg_scalars):
    """Training mode of the declared FPN tube-head extension (the design of the reference's dead
    FPN3D.py:232-330 + tube rois on the 2-MLP head, head_builder.py:29-33, + the 3D keypoint head): per-level losses with
    the scaling of FPN.py:282-321, box / keypoint losses as in the 3D branch of model_builder.py (:481-494 with
    1/time_dim on the box regression, :873-889)."""
    net = Net(weights, opts)
    net.body(torch.from_numpy(data))
    pyr = net.fpn()                                   # [P6, P5, P4, P3, P2], (1, C, T, H, W)
    o = opts
    T, kt = o['num_frames_mid'], o['kt_rpn']
    ng = cfg_scalars['num_gpus']
    losses = {}
    for lvl in range(2, 7):
        x = pyr[6 - lvl]
        h = F.relu(F.conv3d(x, weights['conv_rpn_fpn2_w'], weights['conv_rpn_fpn2_b'], stride=1, padding=(kt // 2, 1, 1)))
        N, C, Tt, H, W = h.shape
        h2 = h.permute(0, 2, 1, 3, 4).reshape(N, Tt * C, H, W)          # channel index t*C + c (detector.py:480-491)
        logits = net.conv2d(h2, 'rpn_cls_logits_fpn2', 1)
        deltas = net.conv2d(h2, 'rpn_bbox_pred_fpn2', 1)
        lab = torch.from_numpy(labels['rpn_labels_int32_wide_fpn%d' % lvl][:, :, :H, :W])
        valid = (lab >= 0).float()
        ce = F.binary_cross_entropy_with_logits(logits, lab.clamp(min=0).float(), reduction='none')
        losses['loss_rpn_cls_fpn%d' % lvl] = (ce * valid).sum() / ng / cfg_scalars['rpn_batch'] / cfg_scalars['ims_per_batch']
        t, wi, wo = [torch.from_numpy(labels['rpn_bbox_%s_wide_fpn%d' % (k, lvl)][:, :, :H, :W])
                     for k in ('targets', 'inside_weights', 'outside_weights')]
        losses['loss_rpn_bbox_fpn%d' % lvl] = smooth_l1(deltas, t, wi, wo, 1. / 9.) / N / ng / T
    rois = sampled['rois']
    R = rois.shape[0]
    feat = roi_feat_fpn_tube_torch(pyr[1:], rois, o['frcn_res'], o['frcn_sampling'])     # (R, C, T, 7, 7)
    x = F.relu(net.fc(feat.reshape(R, -1), 'fc6'))
    x = F.relu(net.fc(x, 'fc7'))
    cls_score, bbox_pred = net.fc(x, 'cls_score'), net.fc(x, 'bbox_pred')
    losses['loss_cls'] = F.cross_entropy(cls_score, torch.from_numpy(sampled['labels_int32']).long(), reduction='sum') / R / ng
    losses['loss_bbox'] = smooth_l1(bbox_pred, torch.from_numpy(sampled['bbox_targets']),
                                    torch.from_numpy(sampled['bbox_inside_weights']),
                                    torch.from_numpy(sampled['bbox_outside_weights']), 1.0) / R / ng / T
    kfeat = roi_feat_fpn_tube_torch(pyr[1:], sampled['keypoint_rois'], o['kps_res'], o['kps_sampling'])
    kps = net.kps_head_tube_feat(kfeat)               # (Rk, T*K, M, M)
    Rk, TK, M, _ = kps.shape
    w = torch.from_numpy(sampled['keypoint_weights']).reshape(-1)
    nll = F.cross_entropy(kps.reshape(Rk * TK, M * M), torch.from_numpy(sampled['keypoint_locations_int32']).reshape(-1).long(),
                          reduction='none')
    losses['loss_kps'] = (nll * w).sum() / w.sum() * cfg_scalars['kps_loss_weight'] / ng
    return losses
