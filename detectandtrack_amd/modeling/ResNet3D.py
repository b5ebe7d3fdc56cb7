"""NC(T)HW ResNet bodies and the res5 RoI head — builder mirror of reference lib/modeling/ResNet3D.py.

Same public function names (they are referenced BY NAME from YAML: `MODEL.CONV_BODY:
ResNet3D.add_ResNet18_conv4_body`, model_builder.py:39-49), same blob / parameter names, same stage
arithmetic.  "3D" here means full kT x 3 x 3 kernels with kT = VIDEO.TIME_KERNEL_DIM.BODY in res3..res5 and
kT = 1 in conv1/res2, never a temporal stride (reference ResNet3D.py:258-284).
"""
from detectandtrack_amd.core.config import cfg
from detectandtrack_amd.modeling.common import ConvStageInfo


def _t_stride(stride, time_stride_on):
    return stride if time_stride_on else 1


def bottleneck_transformation(model, blob_in, dim_in, dim_out, stride, prefix, dim_inner, dilation=1, group=1,
                              time_kernel_dim=1, time_stride_on=False):
    """1x1x1 -> kTx3x3 -> 1x1x1, each followed by AffineChannelNd (reference :21-55).  With
    RESNETS.STRIDE_1X1 the spatial stride sits on the first 1x1x1 (MSRA), else on the 3x3."""
    s1, s3 = (stride, 1) if cfg.RESNETS.STRIDE_1X1 else (1, stride)
    kt = time_kernel_dim
    cur = model.ConvAffineNd(blob_in, prefix + '_branch2a', dim_in, dim_inner, kernels=[1, 1, 1],
                             strides=[_t_stride(s1, time_stride_on), s1, s1], pads=2 * [0, 0, 0], inplace=True)
    cur = model.Relu(cur, cur)
    cur = model.ConvAffineNd(cur, prefix + '_branch2b', dim_inner, dim_inner, kernels=[kt, 3, 3],
                             strides=[_t_stride(s3, time_stride_on), s3, s3], pads=2 * [kt // 2, dilation, dilation],
                             dilations=[1, dilation, dilation], group=group, inplace=True)
    cur = model.Relu(cur, cur)
    return model.ConvAffineNd(cur, prefix + '_branch2c', dim_inner, dim_out, kernels=[1, 1, 1], strides=[1, 1, 1],
                              pads=2 * [0, 0, 0], inplace=False)


def basic_transformation(model, blob_in, dim_in, dim_out, stride, prefix, dim_inner, dilation=1, group=1,
                         time_kernel_dim=1, time_stride_on=False):
    """Two kTx3x3 convs, R-18/34 (reference :59-82)."""
    if dim_inner is None:
        dim_inner = dim_out
    kt = time_kernel_dim
    cur = model.ConvAffineNd(blob_in, prefix + '_branch2a', dim_in, dim_inner, kernels=[kt, 3, 3],
                             strides=[_t_stride(stride, time_stride_on), stride, stride], pads=2 * [kt // 2, 1, 1],
                             inplace=True)
    cur = model.Relu(cur, cur)
    return model.ConvAffineNd(cur, prefix + '_branch2b', dim_inner, dim_out, kernels=[kt, 3, 3], strides=[1, 1, 1],
                              pads=2 * [kt // 2, dilation, dilation], dilations=dilation, group=group, inplace=False)


_TRANS = {'bottleneck_transformation': bottleneck_transformation, 'basic_transformation': basic_transformation}


def add_shortcut(model, prefix, blob_in, dim_in, dim_out, stride, time_stride_on):
    """Identity, or strided 1x1x1 projection + affine (reference :89-101)."""
    if dim_in == dim_out:
        return blob_in
    c = model.ConvNd(blob_in, prefix + '_branch1', dim_in, dim_out, [1, 1, 1],
                     strides=[_t_stride(stride, time_stride_on), stride, stride], pads=2 * [0, 0, 0], no_bias=1)
    return model.AffineChannelNd(c, prefix + '_branch1_bn', dim_out=dim_out)


def add_bottleneck_block(stage_id, model, prefix, blob_in, dim_in, dim_out, dim_inner, dilation, stride_init=2,
                         inplace_sum=False, time_kernel_dim=1, time_stride_on=False):
    """transformation + shortcut -> Sum -> Relu (reference :120-154).  The explicit stage_id keeps stride 1 in
    the first stage even for R-18/34 where dim_in == dim_out == 64 (:133-135)."""
    stride = stride_init if (dim_in != dim_out and stage_id != 1 and dilation == 1) else 1
    tr = _TRANS[cfg.RESNETS.TRANS_FUNC](model, blob_in, dim_in, dim_out, stride, prefix, dim_inner,
                                        group=cfg.RESNETS.NUM_GROUPS, dilation=dilation,
                                        time_kernel_dim=time_kernel_dim, time_stride_on=time_stride_on)
    sc = add_shortcut(model, prefix, blob_in, dim_in, dim_out, stride, time_stride_on=time_stride_on)
    s = model.net.Sum([tr, sc], tr if inplace_sum else prefix + '_sum')
    return model.Relu(s, s)


def add_stage(stage_id, model, prefix, blob_in, n, dim_in, dim_out, dim_inner, dilation, stride_init=2,
              time_kernel_dim=1, time_stride_on=False):
    """n blocks; the last block's sum is a named blob (`<prefix>_<n-1>_sum`) because FPN taps it (:210-227)."""
    for i in range(n):
        blob_in = add_bottleneck_block(stage_id, model, '{}_{}'.format(prefix, i), blob_in, dim_in, dim_out, dim_inner,
                                       dilation, stride_init, inplace_sum=i < n - 1,
                                       time_kernel_dim=time_kernel_dim, time_stride_on=time_stride_on)
        dim_in = dim_out
    return blob_in, dim_in


def add_ResNet_convX_body(model, block_counts, freeze_at=2, feat_dims=(64, 256, 512, 1024, 2048)):
    """data -> conv1 [1,7,7]/[1,2,2] -> affine -> relu -> maxpool [1,3,3]/[1,2,2] -> res2..res4(5) (:251-298)."""
    assert freeze_at in [0, 2, 3, 4, 5]
    p = model.ConvNd('data', 'conv1', 3, feat_dims[0], [1, 7, 7], pads=2 * [0, 3, 3], strides=[1, 2, 2], no_bias=1)
    p = model.AffineChannelNd(p, 'res_conv1_bn', dim_out=feat_dims[0], inplace=True)
    p = model.Relu(p, p)
    p = model.MaxPool(p, 'pool1', kernels=[1, 3, 3], pads=2 * [0, 1, 1], strides=[1, 2, 2])
    dim_in = feat_dims[0]
    dim_b = cfg.RESNETS.NUM_GROUPS * cfg.RESNETS.WIDTH_PER_GROUP
    kt, ts = cfg.VIDEO.TIME_KERNEL_DIM.BODY, cfg.VIDEO.TIME_STRIDE_ON
    n1, n2, n3 = block_counts[:3]
    s, dim_in = add_stage(1, model, 'res2', p, n1, dim_in, feat_dims[1], dim_b, 1, time_kernel_dim=1,
                          time_stride_on=False)
    if freeze_at == 2:
        model.StopGradient(s, s)
    s, dim_in = add_stage(2, model, 'res3', s, n2, dim_in, feat_dims[2], dim_b * 2, 1, time_kernel_dim=kt,
                          time_stride_on=ts)
    if freeze_at == 3:
        model.StopGradient(s, s)
    s, dim_in = add_stage(3, model, 'res4', s, n3, dim_in, feat_dims[3], dim_b * 4, 1, time_kernel_dim=kt,
                          time_stride_on=ts)
    if freeze_at == 4:
        model.StopGradient(s, s)
    if len(block_counts) == 4:
        s, dim_in = add_stage(4, model, 'res5', s, block_counts[3], dim_in, feat_dims[4], dim_b * 8,
                              cfg.MODEL.DILATION, time_kernel_dim=kt, time_stride_on=ts)
        if freeze_at == 5:
            model.StopGradient(s, s)
        return s, dim_in, 1. / 32. * cfg.MODEL.DILATION
    return s, dim_in, 1. / 16.


def add_ResNet_roi_conv5_head(model, blob_in, dim_in, spatial_scale, block_counts=3, dim_out=2048):
    """RoIAlign on tube rois -> res5 stage per RoI (kT = 1) -> mean over H,W, T kept (:301-327)."""
    model.RoIFeatureTransform(blob_in, 'pool5', blob_rois='rois', method=cfg.FAST_RCNN.ROI_XFORM_METHOD,
                              resolution=cfg.FAST_RCNN.ROI_XFORM_RESOLUTION,
                              sampling_ratio=cfg.FAST_RCNN.ROI_XFORM_SAMPLING_RATIO, spatial_scale=spatial_scale)
    dim_b = cfg.RESNETS.NUM_GROUPS * cfg.RESNETS.WIDTH_PER_GROUP
    stride_init = int(cfg.FAST_RCNN.ROI_XFORM_RESOLUTION / 7)
    s, dim_in = add_stage(4, model, 'res5', 'pool5', block_counts, dim_in, dim_out, dim_b * 8, 1, stride_init)
    s = model.SpatialMean(s, 'res5_pool')
    return s, dim_out, spatial_scale


# ---- named architectures (YAML entry points, :333-397) -------------------------------------------------------
def _body(model, counts, trans, dims=None):
    cfg.RESNETS.TRANS_FUNC = trans  # the reference mutates cfg here too (:335)
    if dims is None:
        return add_ResNet_convX_body(model, counts, freeze_at=2)
    return add_ResNet_convX_body(model, counts, freeze_at=2, feat_dims=dims)


def add_ResNet18_conv4_body(model):
    return _body(model, (2, 2, 2), 'basic_transformation', (64, 64, 128, 256))


def add_ResNet18_conv5_body(model):
    return _body(model, (2, 2, 2, 2), 'basic_transformation', (64, 64, 128, 256, 512))


def add_ResNet34_conv4_body(model):
    return _body(model, (3, 4, 6), 'basic_transformation', (64, 64, 128, 256))


def add_ResNet34_conv5_body(model):
    return _body(model, (3, 4, 6, 3), 'basic_transformation', (64, 64, 128, 256, 512))


def add_ResNet50_conv4_body(model):
    return _body(model, (3, 4, 6), 'bottleneck_transformation')


def add_ResNet50_conv5_body(model):
    return _body(model, (3, 4, 6, 3), 'bottleneck_transformation')


def add_ResNet101_conv4_body(model):
    return _body(model, (3, 4, 23), 'bottleneck_transformation')


def add_ResNet101_conv5_body(model):
    return _body(model, (3, 4, 23, 3), 'bottleneck_transformation')


def add_ResNet152_conv5_body(model):
    return _body(model, (3, 8, 36, 3), 'bottleneck_transformation')


def add_ResNet18_roi_conv5_head(*args, **kwargs):
    kwargs.update(dim_out=512, block_counts=2)
    return add_ResNet_roi_conv5_head(*args, **kwargs)


def add_ResNet34_roi_conv5_head(*args, **kwargs):
    kwargs.update(dim_out=512, block_counts=3)
    return add_ResNet_roi_conv5_head(*args, **kwargs)


# ---- stage info for FPN (:401-433) ---------------------------------------------------------------------------------
def _stage_info(last, dims):
    return ConvStageInfo(blobs=tuple('res%d_%d_sum' % (5 - i, n) for i, n in enumerate(last)), dims=dims,
                         spatial_scales=(1. / 32., 1. / 16., 1. / 8., 1. / 4.))


def stage_info_ResNet18_conv5():
    return _stage_info((1, 1, 1, 1), (512, 256, 128, 64))


def stage_info_ResNet34_conv5():
    return _stage_info((2, 5, 3, 2), (512, 256, 128, 64))


def stage_info_ResNet50_conv5():
    return _stage_info((2, 5, 3, 2), (2048, 1024, 512, 256))


def stage_info_ResNet101_conv5():
    return _stage_info((2, 22, 3, 2), (2048, 1024, 512, 256))


def stage_info_ResNet152_conv5():
    return _stage_info((2, 35, 7, 2), (2048, 1024, 512, 256))
