"""Box head — builder mirror of reference lib/modeling/head_builder.py:17-38."""
from detectandtrack_amd.core.config import cfg


def add_roi_2mlp_head(model, blob_in, dim_in, spatial_scale):
    """RoIAlign -> FC -> ReLU -> FC -> ReLU.  fc6 consumes T*C*res*res inputs when the head is 3D (:29-33)."""
    hidden = cfg.FAST_RCNN.MLP_HEAD_DIM
    res = cfg.FAST_RCNN.ROI_XFORM_RESOLUTION
    feat = model.RoIFeatureTransform(blob_in, 'roi_feat', blob_rois='rois', method=cfg.FAST_RCNN.ROI_XFORM_METHOD,
                                     resolution=res, sampling_ratio=cfg.FAST_RCNN.ROI_XFORM_SAMPLING_RATIO,
                                     spatial_scale=spatial_scale)
    t = cfg.VIDEO.NUM_FRAMES_MID if cfg.MODEL.VIDEO_ON and cfg.VIDEO.BODY_HEAD_LINK == '' else 1
    model.frcn_head_is_fc = True    # the outputs that follow are FCs on a 2-D blob (model_builder.add_fast_rcnn_outputs)
    model.FC(feat, 'fc6', t * dim_in * res * res, hidden)
    model.Relu('fc6', 'fc6')
    model.FC('fc6', 'fc7', hidden, hidden)
    model.Relu('fc7', 'fc7')
    return 'fc7', hidden, spatial_scale
