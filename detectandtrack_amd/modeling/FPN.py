"""2D FPN body + FPN-RPN heads + RoI->level mapping — builder mirror of reference lib/modeling/FPN.py.

The 2D body is the 3D one with T = 1 / kT = 1 (same blob names), so it delegates to FPN3D with 2D ResNet
bodies.  `add_fpn_rpn_outputs` (:205-279) is the functional RPN head used by every shipped FPN config,
including the video one (`BODY_HEAD_LINK: slice-center`, configs/video/2d_best/*.yaml).
"""
import numpy as np

from detectandtrack_amd.core.config import cfg
from detectandtrack_amd.modeling.generate_anchors import generate_anchors
import detectandtrack_amd.modeling.FPN3D as FPN3D
import detectandtrack_amd.modeling.ResNet as ResNet
import detectandtrack_amd.utils.boxes as box_utils

LOWEST_LVL = FPN3D.LOWEST_LVL
HIGHEST_LVL = FPN3D.HIGHEST_LVL
get_min_max_levels = FPN3D.get_min_max_levels
add_fpn = FPN3D.add_fpn
add_topdown_lateral_module = FPN3D.add_topdown_lateral_module


def add_fpn_ResNet50_conv5_body(model):
    return FPN3D._onto(model, ResNet.add_ResNet50_conv5_body, ResNet.stage_info_ResNet50_conv5)


def add_fpn_ResNet50_conv5_P2only_body(model):
    return FPN3D._onto(model, ResNet.add_ResNet50_conv5_body, ResNet.stage_info_ResNet50_conv5, True)


def add_fpn_ResNet101_conv5_body(model):
    return FPN3D._onto(model, ResNet.add_ResNet101_conv5_body, ResNet.stage_info_ResNet101_conv5)


def add_fpn_ResNet101_conv5_P2only_body(model):
    return FPN3D._onto(model, ResNet.add_ResNet101_conv5_body, ResNet.stage_info_ResNet101_conv5, True)


def add_fpn_ResNet152_conv5_body(model):
    return FPN3D._onto(model, ResNet.add_ResNet152_conv5_body, ResNet.stage_info_ResNet152_conv5)


def add_fpn_rpn_outputs(model, blobs_in, dim_in, spatial_scales, time_dim=1):
    """Per level: 3x3 conv + ReLU, 1x1 objectness (A), 1x1 deltas (4A); levels > k_min share level k_min's
    parameters; Sigmoid + GenerateProposals per level (reference :205-279)."""
    A = len(cfg.FPN.RPN_ASPECT_RATIOS)
    k_max, k_min = cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.RPN_MIN_LEVEL
    assert len(blobs_in) == k_max - k_min + 1
    g, z = ('GaussianFill', {'std': 0.01}), ('ConstantFill', {'value': 0.})
    first = str(k_min)
    for lvl in range(k_min, k_max + 1):
        bl_in, sc, s = blobs_in[k_max - lvl], spatial_scales[k_max - lvl], str(lvl)
        if lvl == k_min:
            h = model.Conv(bl_in, 'conv_rpn_fpn' + s, dim_in, dim_in, 3, pad=1, stride=1, weight_init=g, bias_init=z)
            model.Relu(h, h)
            lg = model.Conv(h, 'rpn_cls_logits_fpn' + s, dim_in, A, 1, pad=0, stride=1, weight_init=g, bias_init=z)
            bp = model.Conv(h, 'rpn_bbox_pred_fpn' + s, dim_in, 4 * A, 1, pad=0, stride=1, weight_init=g, bias_init=z)
        else:
            h = model.ConvShared(bl_in, 'conv_rpn_fpn' + s, dim_in, dim_in, 3, pad=1, stride=1,
                                 weight='conv_rpn_fpn' + first + '_w', bias='conv_rpn_fpn' + first + '_b')
            model.Relu(h, h)
            lg = model.ConvShared(h, 'rpn_cls_logits_fpn' + s, dim_in, A, 1, pad=0, stride=1,
                                  weight='rpn_cls_logits_fpn' + first + '_w', bias='rpn_cls_logits_fpn' + first + '_b')
            bp = model.ConvShared(h, 'rpn_bbox_pred_fpn' + s, dim_in, 4 * A, 1, pad=0, stride=1,
                                  weight='rpn_bbox_pred_fpn' + first + '_w', bias='rpn_bbox_pred_fpn' + first + '_b')
        if not model.train or cfg.MODEL.FASTER_RCNN:
            anchors = generate_anchors(stride=2. ** lvl, sizes=(cfg.FPN.RPN_ANCHOR_START_SIZE * 2. ** (lvl - k_min),),
                                       aspect_ratios=cfg.FPN.RPN_ASPECT_RATIOS, time_dim=1)
            probs = model.net.Sigmoid(lg, 'rpn_cls_probs_fpn' + s)
            model.GenerateProposals([probs, bp, 'im_info'], ['rpn_rois_fpn' + s, 'rpn_roi_probs_fpn' + s],
                                    anchors=anchors, spatial_scale=sc)


def add_fpn_rpn_losses(model, time_dim=1):
    """:282-321 (shared with FPN3D): per level SigmoidCrossEntropyLoss on the objectness logits (normalize = 0, scaled by
    1 / NUM_GPUS / RPN_BATCH_SIZE_PER_IM / IMS_PER_BATCH) and SmoothL1Loss (beta 1/9) on the deltas; the full-sized
    ("wide") label arrays of the data loader are narrowed to the level's H x W inside the loss kernel."""
    from detectandtrack_amd.modeling.detector import Op
    for lvl in range(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.RPN_MAX_LEVEL + 1):
        s = str(lvl)
        model.net.add(Op('RpnLoss',
                         ['rpn_cls_logits_fpn' + s, 'rpn_bbox_pred_fpn' + s, 'rpn_labels_int32_wide_fpn' + s,
                          'rpn_bbox_targets_wide_fpn' + s, 'rpn_bbox_inside_weights_wide_fpn' + s,
                          'rpn_bbox_outside_weights_wide_fpn' + s],
                         ['loss_rpn_cls_fpn' + s, 'loss_rpn_bbox_fpn' + s],
                         cls_scale=1. / cfg.NUM_GPUS / cfg.TRAIN.RPN_BATCH_SIZE_PER_IM / cfg.TRAIN.IMS_PER_BATCH,
                         normalize=0, beta=1. / 9., bbox_scale=1. / cfg.NUM_GPUS / time_dim))
        model.losses = sorted(set(model.losses + ['loss_rpn_cls_fpn' + s, 'loss_rpn_bbox_fpn' + s]))


def map_rois_to_fpn_levels(rois, k_min, k_max):
    """Eqn.(1) of the FPN paper on the (tube-mean) box area (reference :349-360)."""
    s = np.sqrt(box_utils.boxes_area(rois))
    lvls = np.floor(cfg.FPN.ROI_CANONICAL_LEVEL + np.log2(s / cfg.FPN.ROI_CANONICAL_SCALE + 1e-6))
    return np.clip(lvls, k_min, k_max)


def add_multilevel_roi_blobs(blobs, blob_name, rois, lvls, lvl_min, lvl_max, valid_levels=None):
    """Split rois per level and record the permutation that restores the original order (reference :363-381)."""
    if valid_levels is None:
        valid_levels = 1
    order = np.empty((0,))
    for lvl in range(lvl_min, lvl_max + 1):
        idx = np.where(lvls * valid_levels == lvl)[0]
        blobs[blob_name + '_fpn' + str(lvl)] = rois[idx, :]
        order = np.concatenate((order, idx))
    blobs[blob_name + '_idx_restore_int32'] = np.argsort(order, kind='stable').astype(np.int32, copy=False)
