"""Keypoint head trunk — builder mirror of reference lib/modeling/keypoint_rcnn_heads.py:39-73."""
from detectandtrack_amd.core.config import cfg


def add_roi_pose_head_v1convX(model, blob_in, dim_in, spatial_scale, nd=False):
    """RoIAlign(14x14) -> NUM_STACKED_CONVS x (conv kTx3x3 CONV_HEAD_DIM + ReLU)."""
    hidden, k = cfg.KRCNN.CONV_HEAD_DIM, cfg.KRCNN.CONV_HEAD_KERNEL
    cur = model.RoIFeatureTransform(blob_in, '_[pose]_roi_feat', blob_rois='keypoint_rois',
                                    method=cfg.KRCNN.ROI_XFORM_METHOD, resolution=cfg.KRCNN.ROI_XFORM_RESOLUTION,
                                    sampling_ratio=cfg.KRCNN.ROI_XFORM_SAMPLING_RATIO, spatial_scale=spatial_scale)
    init, zero = (cfg.KRCNN.CONV_INIT, {'std': 0.01}), ('ConstantFill', {'value': 0.})
    kt = cfg.VIDEO.TIME_KERNEL_DIM.HEAD_KPS
    for i in range(cfg.KRCNN.NUM_STACKED_CONVS):
        name = 'conv_fcn' + str(i + 1)
        if nd:
            cur = model.ConvNd(cur, name, dim_in, hidden, [kt, k, k], pads=2 * [kt // 2, k // 2, k // 2],
                               strides=[1, 1, 1], weight_init=init, bias_init=zero)
        else:
            cur = model.Conv(cur, name, dim_in, hidden, k, stride=1, pad=k // 2, weight_init=init, bias_init=zero)
        cur = model.Relu(cur, cur)
        dim_in = hidden
    return cur, hidden, spatial_scale


def add_roi_pose_head_v1convX_3d(model, blob_in, dim_in, spatial_scale):
    return add_roi_pose_head_v1convX(model, blob_in, dim_in, spatial_scale, nd=True)
