"""Model assembly — builder mirror of reference lib/modeling/model_builder.py.

`create(cfg.MODEL.TYPE, train)` dispatches BY NAME exactly like the reference (:39-61): the model type and the
body/head functions named in the YAML (`MODEL.CONV_BODY: FPN3D.add_fpn_ResNet18_conv5_body`, ...) are looked up
in this module's globals.  The result is a DetectionModelHelper whose `.net` / `.keypoint_net` /
`.conv_body_net` are recorded op lists executed by `detectandtrack_amd.workspace` on the MI355X.

Inference graphs are complete.  `create(type, train=True)` adds the training pieces of §8 a12: the loss ops (:481-498 / :612-660 /
:873-905), GenerateProposalLabels and StopGradient; gradients, the all-reduce and MomentumSGDUpdate (:908-985) live in
`detectandtrack_amd.training`, the input pipeline in `detectandtrack_amd.roi_data.loader`.  Only the FPN tube-head extension has no
training graph yet (build_generic_fast_rcnn_model raises).
"""
import logging

from detectandtrack_amd.core.config import cfg
from detectandtrack_amd.modeling.detector import DetectionModelHelper, Net
from detectandtrack_amd.modeling.generate_anchors import generate_anchors
import detectandtrack_amd.modeling.FPN as FPN  # noqa  (YAML)
import detectandtrack_amd.modeling.FPN3D as FPN3D  # noqa  (YAML)
import detectandtrack_amd.modeling.ResNet as ResNet  # noqa  (YAML)
import detectandtrack_amd.modeling.ResNet3D as ResNet3D  # noqa  (YAML)
import detectandtrack_amd.modeling.head_builder as head_builder  # noqa  (YAML)
import detectandtrack_amd.modeling.keypoint_rcnn_heads as keypoint_rcnn_heads  # noqa  (YAML)

logger = logging.getLogger(__name__)


def get_func(func_name):
    """'Module.function' -> function object, resolved in this module's globals (:39-49)."""
    try:
        parts = func_name.split('.')
        res = globals()[parts[0]]
        for part in parts[1:]:
            res = getattr(res, part)
        return res
    except Exception:
        logger.error('Failed to find function: {}'.format(func_name))
        raise


def init_model(model_name, train, init_params=None):
    return DetectionModelHelper(name=model_name, train=train, num_classes=cfg.MODEL.NUM_CLASSES,
                                init_params=init_params)


def create(model_name, train=False, init_params=None):
    """:52-61."""
    return get_func(model_name)(init_model(model_name, train, init_params))


# ---- model types (:81-175) -------------------------------------------------------------------------------------
def faster_rcnn(model):
    assert cfg.MODEL.FASTER_RCNN
    return build_generic_fast_rcnn_model(model, get_func(cfg.MODEL.CONV_BODY), get_func(cfg.MODEL.ROI_HEAD))


def keypoint_rcnn(model):
    return build_generic_fast_rcnn_model(model, get_func(cfg.MODEL.CONV_BODY), get_func(cfg.MODEL.ROI_HEAD),
                                         add_roi_keypoint_head_func=get_func(cfg.KRCNN.ROI_KEYPOINTS_HEAD))


def keypoint_rcnn_frozen_features(model):
    return build_generic_fast_rcnn_model(model, get_func(cfg.MODEL.CONV_BODY), get_func(cfg.MODEL.ROI_HEAD),
                                         add_roi_keypoint_head_func=get_func(cfg.KRCNN.ROI_KEYPOINTS_HEAD),
                                         freeze_conv_body=True)


def mask_rcnn(model):
    raise NotImplementedError('mask branch is out of the hot-path scope: it raises for tubes in the reference '
                              '(core/test.py:916-917)')


def blob_ref_to_list(b):
    return b if isinstance(b, list) else [b]


def time_pool_blobs(blob_conv, model, body_head_link):
    """3D body -> 2D head link (:1024-1042): '' keeps T, 'avg' averages T, 'slice-center' takes the key frame."""
    if body_head_link == '':
        return blob_conv
    out = []
    for blob in blob_ref_to_list(blob_conv):
        if body_head_link == 'avg':
            out.append(model.TimePool(blob, None, 'avg'))
        elif body_head_link == 'slice-center':
            out.append(model.SliceKeyFrame(blob, cfg.VIDEO.NUM_FRAMES_MID))
        else:
            raise NotImplementedError('Uknown body-head link {}'.format(body_head_link))
    return out if isinstance(blob_conv, list) else out[0]


def build_generic_fast_rcnn_model(model, add_conv_body_func, add_roi_frcn_head_func, add_roi_mask_head_func=None,
                                  add_roi_keypoint_head_func=None, freeze_conv_body=False):
    """:179-306 (single replica; the reference loops this over NUM_GPUS name scopes for training)."""
    if model.train and not cfg.MODEL.FASTER_RCNN:
        raise NotImplementedError('training graph: end-to-end Faster R-CNN only (pre-computed proposal training needs the dataset layer)')
    blob_conv, dim_conv, spatial_scale_conv = add_conv_body_func(model)
    if cfg.MODEL.VIDEO_ON:
        blob_conv = time_pool_blobs(blob_conv, model, cfg.VIDEO.BODY_HEAD_LINK)
    model.conv_body_net = model.net.Clone('conv_body_net')

    if cfg.MODEL.VIDEO_ON and cfg.VIDEO.BODY_HEAD_LINK == '':
        fpn_lib, head_3d, out_time_dim = FPN3D, True, cfg.VIDEO.NUM_FRAMES_MID
    else:
        fpn_lib, head_3d, out_time_dim = FPN, False, 1

    if cfg.MODEL.FASTER_RCNN:
        if cfg.FPN.FPN_ON:
            fpn_lib.add_fpn_rpn_outputs(model, blob_conv, dim_conv, spatial_scale_conv, time_dim=out_time_dim)
            if model.train:
                FPN.add_fpn_rpn_losses(model, time_dim=out_time_dim)
            model.CollectAndDistributeFpnRpnProposals()
        else:
            add_rpn_outputs(model, blob_conv, dim_conv, spatial_scale_conv, nd=head_3d, time_dim=out_time_dim)
            if model.train:
                add_rpn_losses(model, time_dim=out_time_dim)

    if cfg.FPN.FPN_ON:
        assert cfg.FPN.RPN_MIN_LEVEL == cfg.FPN.ROI_MIN_LEVEL
        n_roi_levels = cfg.FPN.ROI_MAX_LEVEL - cfg.FPN.ROI_MIN_LEVEL + 1
        blob_conv = blob_conv[-n_roi_levels:]
        spatial_scale_conv = spatial_scale_conv[-n_roi_levels:]

    blob_frcn, dim_frcn, _ = add_roi_frcn_head_func(model, blob_conv, dim_conv, spatial_scale_conv)
    add_fast_rcnn_outputs(model, blob_frcn, dim_frcn, is_head_3d=head_3d, time_dim=out_time_dim)
    if model.train:
        add_fast_rcnn_losses(model, time_dim=out_time_dim)

    if cfg.MODEL.MASK_ON:
        raise NotImplementedError('mask branch out of scope (core/test.py:916-917 raises for tubes)')

    if cfg.MODEL.KEYPOINTS_ON:
        n_bbox_ops = len(model.net.ops)
        blob_krcnn, dim_krcnn, _ = add_roi_keypoint_head_func(model, blob_conv, dim_conv, spatial_scale_conv)
        add_heatmap_outputs(model, blob_krcnn, dim_krcnn, time_dim=out_time_dim, is_head_3d=head_3d)
        if model.train:
            # training: the keypoint branch stays in the main net and gets its loss (:269-273)
            add_heatmap_losses(model, time_dim=out_time_dim)
        else:
            # inference: the keypoint branch is its own net, run only on the surviving detections (:264-267, :994-1021)
            model.keypoint_net = Net('keypoint_net', model)
            model.keypoint_net.ops = model.net.ops[n_bbox_ops:]
            model.net.ops = model.net.ops[:n_bbox_ops]
    return model


# ---- losses (:481-494, :873-889; RPN: FPN.py:282-321) -----------------------------------------------------------------
def add_fast_rcnn_losses(model, time_dim=1):
    model.net.add(_op('SoftmaxLoss', ['cls_score', 'labels_int32'], ['cls_prob', 'loss_cls', 'accuracy_cls'],
                      scale=1. / cfg.NUM_GPUS))
    model.net.add(_op('SmoothL1Loss', ['bbox_pred', 'bbox_targets', 'bbox_inside_weights', 'bbox_outside_weights'],
                      ['loss_bbox'], beta=1.0, scale=1. / cfg.NUM_GPUS / time_dim))
    model.losses = sorted(set(model.losses + ['loss_cls', 'loss_bbox']))
    model.metrics = sorted(set(model.metrics + ['accuracy_cls']))


def add_heatmap_losses(model, time_dim=1):
    """kps_score (R, K, M, M) -> (R*K, M*M) rows; SoftmaxWithLoss across SPACE with per-keypoint weights; the loss is
    not scaled by time_dim (:884-887)."""
    model.net.add(_op('KeypointLoss', ['kps_score', 'keypoint_locations_int32', 'keypoint_weights'],
                      ['kps_prob', 'loss_kps'], scale=cfg.KRCNN.LOSS_WEIGHT / cfg.NUM_GPUS))
    model.losses = sorted(set(model.losses + ['loss_kps']))


# ---- Fast R-CNN outputs (:426-478) ----------------------------------------------------------------------------------
def add_fast_rcnn_outputs(model, blob_in, dim, is_head_3d, time_dim=1):
    g01, g001, z = ('GaussianFill', {'std': 0.01}), ('GaussianFill', {'std': 0.001}), ('ConstantFill', {'value': 0.})
    if is_head_3d and getattr(model, 'frcn_head_is_fc', False):
        # DECLARED EXTENSION: tube rois on the FPN 2-MLP head (fc6 over T*C*res*res inputs,
        # head_builder.py:29-33).  The reference would feed the 2-D fc7 into ConvNd here (:427-441), which Caffe2
        # rejects; the outputs are FCs with the tube layout of the 3D branch: K scores, K*T*4 deltas
        # (class-major / frame / xywh, :446-473).
        model.FC(blob_in, 'cls_score', dim, model.num_classes, weight_init=g01, bias_init=z)
        model.Softmax('cls_score', 'cls_prob', engine='CUDNN')
        model.FC(blob_in, 'bbox_pred', dim, model.num_classes * 4 * time_dim, weight_init=g001, bias_init=z)
    elif is_head_3d:
        # 1x1x1 convs on the R x C x T x 1 x 1 head output; class scores averaged over T, box deltas regrouped to
        # class-major / frame / xyxy (:427-473)
        c = model.ConvNd(blob_in, 'cls_score_1', dim, model.num_classes, [1, 1, 1], pads=2 * [0, 0, 0],
                         strides=[1, 1, 1], weight_init=g01, bias_init=z)
        model.TimeMean(c, 'cls_score')
        if not model.train:
            model.Softmax('cls_score', 'cls_prob', engine='CUDNN')
        b = model.ConvNd(blob_in, 'bbox_pred_1', dim, 4 * model.num_classes, [1, 1, 1], pads=2 * [0, 0, 0],
                         strides=[1, 1, 1], weight_init=g01, bias_init=z)
        model.net.add(_op('TubeDeltasToRows', [b], ['bbox_pred']))
    else:
        model.FC(blob_in, 'cls_score', dim, model.num_classes, weight_init=g01, bias_init=z)
        if not model.train:   # training fuses the softmax into the loss for stability (:445-448)
            model.Softmax('cls_score', 'cls_prob', engine='CUDNN')
        model.FC(blob_in, 'bbox_pred', dim, model.num_classes * 4, weight_init=g001, bias_init=z)


def _op(type_, ins, outs, **kw):
    from detectandtrack_amd.modeling.detector import Op
    return Op(type_, [str(i) for i in ins], [str(o) for o in outs], **kw)


# ---- single-level (C4) RPN, 2D or tube (:500-609) ----------------------------------------------------------------------
def add_rpn_outputs(model, blob_in, dim_in, spatial_scale, nd=False, time_dim=1):
    anchors = generate_anchors(stride=1. / spatial_scale, sizes=cfg.RPN.SIZES, aspect_ratios=cfg.RPN.ASPECT_RATIOS,
                               time_dim=time_dim)
    A = anchors.shape[0]
    g, z = ('GaussianFill', {'std': 0.01}), ('ConstantFill', {'value': 0.})
    if nd:
        kt = cfg.VIDEO.TIME_KERNEL_DIM.HEAD_RPN
        model.ConvNd(blob_in, 'conv_rpn', dim_in, dim_in, [kt, 3, 3], pads=2 * [kt // 2, 1, 1], strides=[1, 1, 1],
                     weight_init=g, bias_init=z)
        model.Relu('conv_rpn', 'conv_rpn')
        lg = model.ConvNd('conv_rpn', 'rpn_cls_logits_1', dim_in, A, [1, 1, 1], pads=2 * [0, 0, 0], strides=[1, 1, 1],
                          weight_init=g, bias_init=z)
        model.TimeMean(lg, 'rpn_cls_logits')                       # TimePool 'avg' (:532)
        # deltas: 4A channels per frame; GenerateProposals reads them as (anchor, frame, xywh) (:545-563)
        d = model.ConvNd('conv_rpn', 'rpn_bbox_pred_1', dim_in, 4 * A, [1, 1, 1], pads=2 * [0, 0, 0],
                         strides=[1, 1, 1], weight_init=g, bias_init=z)
        model.net.add(_op('RpnDeltasPerFrame', [d], ['rpn_bbox_pred']))
    else:
        model.Conv(blob_in, 'conv_rpn', dim_in, dim_in, 3, pad=1, stride=1, weight_init=g, bias_init=z)
        model.Relu('conv_rpn', 'conv_rpn')
        model.Conv('conv_rpn', 'rpn_cls_logits', dim_in, A, 1, pad=0, stride=1, weight_init=g, bias_init=z)
        model.Conv('conv_rpn', 'rpn_bbox_pred', dim_in, 4 * A, 1, pad=0, stride=1, weight_init=g, bias_init=z)
    if cfg.MODEL.FASTER_RCNN or (cfg.MODEL.RPN_ONLY and not model.train):
        model.net.Sigmoid('rpn_cls_logits', 'rpn_cls_probs')
        model.GenerateProposals(['rpn_cls_probs', 'rpn_bbox_pred', 'im_info'], ['rpn_rois', 'rpn_roi_probs'],
                                anchors=anchors, spatial_scale=spatial_scale)
    if cfg.MODEL.FASTER_RCNN and not model.train:
        model.net.Alias('rpn_rois', 'rois')
    elif cfg.MODEL.FASTER_RCNN:
        # training: sample labelled rois from the in-network proposals (ops/generate_proposal_labels.py:23-37)
        outs = ['rois', 'labels_int32', 'bbox_targets', 'bbox_inside_weights', 'bbox_outside_weights']
        if cfg.MODEL.KEYPOINTS_ON:
            outs += ['keypoint_rois', 'keypoint_locations_int32', 'keypoint_weights', 'keypoint_loss_normalizer']
        model.net.add(_op('GenerateProposalLabels', ['rpn_rois', 'roidb', 'im_info'], outs))


def add_rpn_losses(model, time_dim=1):
    """:612-636 (single-level RPN, 2D or tube): SigmoidCrossEntropyLoss (normalised by the number of non-ignored anchors) and
    SmoothL1Loss (beta 1/9) on the narrowed 'wide' label arrays."""
    model.net.add(_op('RpnLoss', ['rpn_cls_logits', 'rpn_bbox_pred', 'rpn_labels_int32_wide', 'rpn_bbox_targets_wide',
                                  'rpn_bbox_inside_weights_wide', 'rpn_bbox_outside_weights_wide'],
                      ['loss_rpn_cls', 'loss_rpn_bbox'], cls_scale=1. / cfg.NUM_GPUS, normalize=1, beta=1. / 9.,
                      bbox_scale=1. / cfg.NUM_GPUS / time_dim))
    model.losses = sorted(set(model.losses + ['loss_rpn_cls', 'loss_rpn_bbox']))


# ---- keypoint heatmap outputs (:755-870) ----------------------------------------------------------------------------------
def add_heatmap_outputs(model, blob_in, dim, time_dim, is_head_3d):
    """Trunk output -> ConvTranspose k4 s2 (K maps, 2x) -> fixed bilinear ConvTranspose (UP_SCALE x).  With a 3D
    head the deconvs run per frame, either with shared weights (NO_3D_DECONV_TIME_TO_CH: time -> batch, :760-764; what the
    shipped 3D configs set) or -- the reference default -- with one weight block per frame (time -> channels + group = T,
    :765-767, :848-856); either way the K maps of the T frames end up channel-concatenated as t*K + k (:864-868)."""
    if is_head_3d and cfg.KRCNN.USE_3D_DECONV:
        raise NotImplementedError('ConvTranspose3D is unavailable in the reference too (utils/net.py:55-56)')
    if cfg.KRCNN.USE_DECONV:
        raise NotImplementedError('KRCNN.USE_DECONV intermediate deconv is not used by any shipped config')
    if cfg.KRCNN.UP_SCALE == 1 or not cfg.KRCNN.USE_DECONV_OUTPUT:
        raise NotImplementedError('shipped configs use USE_DECONV_OUTPUT True with UP_SCALE 2')
    K = cfg.KRCNN.NUM_KEYPOINTS
    pad = int(cfg.KRCNN.DECONV_KERNEL / 2 - 1)
    winit, binit = (cfg.KRCNN.CONV_INIT, {'std': 0.001}), ('ConstantFill', {'value': 0.})
    if is_head_3d and not cfg.KRCNN.NO_3D_DECONV_TIME_TO_CH:
        # the reference DEFAULT (core/config.py:472, :765-767 / :848-868): time -> channels (index t*C + c), ONE grouped deconv with
        # group = time_dim -- frame t has its own [C, K, 4, 4] weight block and its own K biases -- and the fixed bilinear deconv on
        # the T*K maps; `kps_score` comes out as (R, T*K, M, M) directly
        blob_in = model.MoveTimeToChannelDim(blob_in, None)
        low = model.ConvTranspose(blob_in, 'kps_score_lowres', dim * time_dim, K * time_dim, cfg.KRCNN.DECONV_KERNEL, pad=pad, stride=2,
                                  group=time_dim, weight_init=winit, bias_init=binit)
        return model.BilinearInterpolation(low, 'kps_score', K * time_dim, K * time_dim, cfg.KRCNN.UP_SCALE)
    if is_head_3d:
        original_time_dim = model.GetTemporalDim(blob_in)
        blob_in = model.MoveTimeToBatchDim(blob_in, None)
    low = model.ConvTranspose(blob_in, 'kps_score_lowres', dim, K, cfg.KRCNN.DECONV_KERNEL, pad=pad, stride=2,
                              weight_init=winit, bias_init=binit)
    if not is_head_3d:
        return model.BilinearInterpolation(low, 'kps_score', K, K, cfg.KRCNN.UP_SCALE)
    # :858-868: the up-sampled maps of the R*T frames, then batch -> time and time -> channels: (R*T, K, M, M) -> (R, K, T, M, M) ->
    # (R, T*K, M, M).  The executor's output kernel writes that last layout directly; the two moves are views of it.  (Like the reference,
    # the function returns the pre-move blob.)
    blob_out = model.BilinearInterpolation(low, 'kps_score_prefinal', K, K, cfg.KRCNN.UP_SCALE)
    model.MoveTimeToBatchDimInverse('kps_score_prefinal', 'kps_score_prefinal2', original_time_dim)
    model.MoveTimeToChannelDim('kps_score_prefinal2', 'kps_score')
    return blob_out
