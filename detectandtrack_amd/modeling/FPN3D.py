"""3D feature pyramid — builder mirror of reference lib/modeling/FPN3D.py.

Body part (add_fpn, :109-222) is functional in the reference and reproduced here: 1x1x1 laterals, spatial-only
nearest 2x top-down (time untouched), kT x 3 x 3 post-hoc convs, P6 by stride-2 sub-sampling of P5.  The
reference's 3D FPN *RPN head* raises NotImplementedError (:225-228); `add_fpn_rpn_outputs` below
implements the design its dead code describes and is a DECLARED EXTENSION (no reference behaviour to match).
"""
from detectandtrack_amd.core.config import cfg
from detectandtrack_amd.modeling.generate_anchors import generate_anchors
import detectandtrack_amd.modeling.ResNet3D as ResNet

LOWEST_LVL = 2
HIGHEST_LVL = 5


def _onto(model, body, info, P2only=False):
    body(model)
    blobs, dim, scales = add_fpn(model, info())
    if P2only:
        return blobs[-1], dim, scales[-1]
    return blobs, dim, scales


def add_fpn_ResNet18_conv5_body(model):
    return _onto(model, ResNet.add_ResNet18_conv5_body, ResNet.stage_info_ResNet18_conv5)


def add_fpn_ResNet34_conv5_body(model):
    return _onto(model, ResNet.add_ResNet34_conv5_body, ResNet.stage_info_ResNet34_conv5)


def add_fpn_ResNet50_conv5_body(model):
    return _onto(model, ResNet.add_ResNet50_conv5_body, ResNet.stage_info_ResNet50_conv5)


def add_fpn_ResNet50_conv5_P2only_body(model):
    return _onto(model, ResNet.add_ResNet50_conv5_body, ResNet.stage_info_ResNet50_conv5, True)


def add_fpn_ResNet101_conv5_body(model):
    return _onto(model, ResNet.add_ResNet101_conv5_body, ResNet.stage_info_ResNet101_conv5)


def add_fpn_ResNet101_conv5_P2only_body(model):
    return _onto(model, ResNet.add_ResNet101_conv5_body, ResNet.stage_info_ResNet101_conv5, True)


def add_fpn_ResNet152_conv5_body(model):
    return _onto(model, ResNet.add_ResNet152_conv5_body, ResNet.stage_info_ResNet152_conv5)


def add_fpn_ResNet152_conv5_P2only_body(model):
    return _onto(model, ResNet.add_ResNet152_conv5_body, ResNet.stage_info_ResNet152_conv5, True)


def add_fpn_generic_onto_body(model, conv_body_func, stage_info_func, P2only=False):
    return _onto(model, conv_body_func, stage_info_func, P2only)


def get_min_max_levels():
    """reference :75-89."""
    lo, hi = LOWEST_LVL, HIGHEST_LVL
    if cfg.FPN.MULTILEVEL_RPN and not cfg.FPN.MULTILEVEL_ROIS:
        hi, lo = cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.RPN_MIN_LEVEL
    if not cfg.FPN.MULTILEVEL_RPN and cfg.FPN.MULTILEVEL_ROIS:
        hi, lo = cfg.FPN.ROI_MAX_LEVEL, cfg.FPN.ROI_MIN_LEVEL
    if cfg.FPN.MULTILEVEL_RPN and cfg.FPN.MULTILEVEL_ROIS:
        hi = max(cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.ROI_MAX_LEVEL)
        lo = min(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.ROI_MIN_LEVEL)
    return lo, hi


def add_fpn(model, stage_info):
    """reference :109-183.  Returns blobs coarse -> fine ([P6,] P5, P4, P3, P2), dim, scales."""
    dim = cfg.FPN.DIM
    kt = cfg.VIDEO.TIME_KERNEL_DIM.BODY
    min_level, max_level = get_min_max_levels()
    xavier, zero = ('XavierFill', {}), ('ConstantFill', {'value': 0.})
    n = len(stage_info.blobs)
    model.ConvNd(stage_info.blobs[0], 'fpn_inner_' + stage_info.blobs[0], stage_info.dims[0], dim, [1, 1, 1],
                 pads=2 * [0, 0, 0], strides=[1, 1, 1], weight_init=xavier, bias_init=zero)
    for i in range(n - 1 - (min_level - LOWEST_LVL)):
        add_topdown_lateral_module(model, 'fpn_inner_' + stage_info.blobs[i], stage_info.blobs[i + 1],
                                   'fpn_inner_' + stage_info.blobs[i + 1], dim, stage_info.dims[i + 1])
    blobs, scales = [], []
    for i in range(n - (min_level - LOWEST_LVL)):
        blobs.append(model.ConvNd('fpn_inner_' + stage_info.blobs[i], 'fpn_' + stage_info.blobs[i], dim, dim,
                                  [kt, 3, 3], pads=2 * [kt // 2, 1, 1], strides=[1, 1, 1], weight_init=xavier,
                                  bias_init=zero))
        scales.append(stage_info.spatial_scales[i])
    if not cfg.FPN.EXTRA_CONV_LEVELS and max_level == HIGHEST_LVL + 1:
        # P6 = MaxPool(kernel 1, stride [1,2,2]) of P5 (:155-164)
        p6 = model.MaxPool(blobs[0], blobs[0] + '_subsampled_2x', kernels=[1, 1, 1], pads=2 * [0, 0, 0],
                           strides=[1, 2, 2])
        blobs.insert(0, p6)
        scales.insert(0, scales[0] * 0.5)
    if cfg.FPN.EXTRA_CONV_LEVELS and max_level > HIGHEST_LVL:
        cur, dim_in = stage_info.blobs[0], stage_info.dims[0]
        for lvl in range(HIGHEST_LVL + 1, max_level + 1):
            if lvl > HIGHEST_LVL + 1:
                cur = model.Relu(cur, cur)
            cur = model.ConvNd(cur, 'fpn_' + str(lvl), dim_in, dim, [kt, 3, 3], pads=2 * [kt // 2, 1, 1],
                               strides=[1, 2, 2], weight_init=xavier, bias_init=zero)
            dim_in = dim
            blobs.insert(0, cur)
            scales.insert(0, scales[0] * 0.5)
    return blobs, dim, scales


def add_topdown_lateral_module(model, fpn_top, fpn_lateral, fpn_bottom, dim_top, dim_lateral):
    """lateral 1x1x1 + nearest-2x (H, W only) of the coarser level, summed (:186-222).  The reference moves time
    into channels around a 2D UpsampleNearest; in NDHWC the upsample is an index map fused into the lateral conv."""
    if cfg.VIDEO.TIME_STRIDE_ON:
        raise NotImplementedError('temporal up-sampling is not defined by the reference either (:199-203)')
    lat = model.ConvNd(fpn_lateral, fpn_bottom if cfg.FPN.INPLACE_LATERAL else fpn_bottom + '_lateral',
                       dim_lateral, dim_top, [1, 1, 1], pads=2 * [0, 0, 0], strides=[1, 1, 1],
                       weight_init=(('ConstantFill', {'value': 0.}) if cfg.FPN.ZERO_INIT_LATERAL else ('XavierFill', {})),
                       bias_init=('ConstantFill', {'value': 0.}))
    td = model.net.UpsampleNearest(fpn_top, fpn_bottom + '_topdown', scale=2)
    model.net.Sum([lat, td], fpn_bottom)


def add_fpn_rpn_outputs(model, blobs_in, dim_in, spatial_scales, time_dim):
    """DECLARED EXTENSION — tube RPN on the 3D pyramid.  The reference raises before its own
    body (:235 'Redo bbox_targets like in model_builder.py'); the design below IS that dead body (:232-330): per level
    a kT x 3 x 3 conv + ReLU, time moved into channels (channel index t*C + c), then 2D 1x1 heads over dim*T inputs:
    objectness (A channels) and tube deltas (4*T*A channels, anchor-major / frame / xywh, what GenerateProposals
    reads with tube anchors).  Heads share weights across levels from level k_min.  The reference's unused
    'rpn_vis_cls_logits' head (:266-272, 'TODO need to use this in future') is not built."""
    A = len(cfg.FPN.RPN_ASPECT_RATIOS)
    k_max, k_min = cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.RPN_MIN_LEVEL
    assert len(blobs_in) == k_max - k_min + 1
    kt = cfg.VIDEO.TIME_KERNEL_DIM.HEAD_RPN
    g, z = ('GaussianFill', {'std': 0.01}), ('ConstantFill', {'value': 0.})
    smin = str(k_min)
    for lvl in range(k_min, k_max + 1):
        bl_in, sc, s = blobs_in[k_max - lvl], spatial_scales[k_max - lvl], str(lvl)

        def shared(n):
            return {} if lvl == k_min else dict(weight=n + smin + '_w', bias=n + smin + '_b')
        h = model.ConvNd(bl_in, 'conv_rpn_fpn' + s, dim_in, dim_in, [kt, 3, 3], pads=2 * [kt // 2, 1, 1],
                         strides=[1, 1, 1], weight_init=g, bias_init=z, **shared('conv_rpn_fpn'))
        model.Relu(h, h)
        h2 = model.MoveTimeToChannelDim(h, 'conv_rpn_timepooled_fpn' + s)
        lg = model.Conv(h2, 'rpn_cls_logits_fpn' + s, dim_in * time_dim, A, 1, pad=0, stride=1, weight_init=g,
                        bias_init=z, **shared('rpn_cls_logits_fpn'))
        bp = model.Conv(h2, 'rpn_bbox_pred_fpn' + s, dim_in * time_dim, 4 * time_dim * A, 1, pad=0, stride=1,
                        weight_init=g, bias_init=z, **shared('rpn_bbox_pred_fpn'))
        anchors = generate_anchors(stride=2. ** lvl, sizes=(cfg.FPN.RPN_ANCHOR_START_SIZE * 2. ** (lvl - k_min),),
                                   aspect_ratios=cfg.FPN.RPN_ASPECT_RATIOS, time_dim=time_dim)
        probs = model.net.Sigmoid(lg, 'rpn_cls_probs_fpn' + s)
        model.GenerateProposals([probs, bp, 'im_info'], ['rpn_rois_fpn' + s, 'rpn_roi_probs_fpn' + s],
                                anchors=anchors, spatial_scale=sc)
