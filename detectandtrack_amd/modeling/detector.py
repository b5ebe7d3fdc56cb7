"""DetectionModelHelper — the builder surface of reference lib/modeling/detector.py, re-targeted.

The reference helper appends Caffe2 operators to a NetDef that the Caffe2 runtime later
executes op by op (ConvNd, AffineChannelNd, Relu, Sum, ... each a full-tensor pass).  Here the
same builder calls RECORD a small op list, and fold at record time everything the MI355X conv
kernel does in its epilogue:

    ConvNd -> AffineChannelNd -> [Sum with shortcut] -> Relu          ==> one 'Conv' op
    ConvNd(lateral) + UpsampleNearest(top) -> Sum  (FPN3D.py:207-222)  ==> one 'Conv' op, res_mode 2

so the op list that `detectandtrack_amd.workspace` executes maps ~1:1 onto HIP kernel launches.
Builder-visible behaviour kept from the reference: blob/param naming, `params`/`weights`/`biases`
bookkeeping (detector.py:57-65, 89-115), init specs, the time<->batch/channel helper names, Python-op
wrappers (GenerateProposals :144, CollectAndDistributeFpnRpnProposals :168, RoIFeatureTransform :256).
"""
import numpy as np

from detectandtrack_amd.core.config import cfg


class Op(object):
    __slots__ = ('type', 'inputs', 'outputs', 'args')

    def __init__(self, type_, inputs, outputs, **args):
        self.type = type_
        self.inputs = list(inputs)
        self.outputs = list(outputs)
        self.args = args

    def __repr__(self):
        return '%s(%s -> %s)' % (self.type, ','.join(self.inputs), ','.join(self.outputs))


class Net(object):
    """Recorded op list (stands in for caffe2.python.core.Net)."""

    def __init__(self, name, helper=None):
        self.name = name
        self.ops = []
        self._helper = helper
        self._next = 0

    def Proto(self):
        return self

    def NextName(self):
        self._next += 1
        return '_auto_%d' % self._next

    def Clone(self, name):
        n = Net(name, self._helper)
        n.ops = list(self.ops)
        return n

    def producer(self, blob):
        for op in reversed(self.ops):
            if blob in op.outputs:
                return op
        return None

    def consumers(self, blob, after_op=None):
        start = 0 if after_op is None else self.ops.index(after_op) + 1
        return [op for op in self.ops[start:] if blob in op.inputs]

    def add(self, op):
        self.ops.append(op)
        return op.outputs[0] if op.outputs else None

    # ---- raw net ops used by the builders (reference: model.net.<Op>) ----
    def Sum(self, blobs_in, blob_out):
        a, b = [str(x) for x in blobs_in]
        blob_out = str(blob_out)
        # fold "conv + shortcut" / "lateral + upsampled top-down" into the conv epilogue
        for x, y in ((a, b), (b, a)):
            pa = self.producer(x)
            if pa is not None and pa.type == 'Conv' and not pa.args['relu'] and pa.args['residual'] is None \
                    and not self.consumers(x, pa):
                py = self.producer(y)
                res, mode = y, 1
                if py is not None and py.type == 'UpsampleNearest2x':
                    res, mode = py.inputs[0], 2
                    if not self.consumers(y, py):
                        self.ops.remove(py)
                self.ops.remove(pa)           # re-append: the residual must already be computed
                pa.args['residual'] = res
                pa.args['res_mode'] = mode
                pa.inputs.append(res)
                pa.outputs = [blob_out]
                self.ops.append(pa)
                return blob_out
        return self.add(Op('Sum', [a, b], [blob_out]))

    def UpsampleNearest(self, blob_in, blob_out, scale=2):
        assert scale == 2, 'only the FPN 2x top-down upsample is used (FPN3D.py:211-212)'
        return self.add(Op('UpsampleNearest2x', [str(blob_in)], [str(blob_out)]))

    def Sigmoid(self, blob_in, blob_out):
        return self.add(Op('Sigmoid', [str(blob_in)], [str(blob_out)]))

    def Alias(self, blob_in, blob_out):
        return self.add(Op('Alias', [str(blob_in)], [str(blob_out)]))

    def BlobIsDefined(self, blob):
        return self.producer(str(blob)) is not None


class DetectionModelHelper(object):
    def __init__(self, name='model', train=False, num_classes=-1, init_params=None):
        self.name = name
        self.train = train
        self.num_classes = num_classes
        self.init_params = train if init_params is None else init_params
        self.net = Net(name, self)
        self.conv_body_net = None
        self.keypoint_net = None
        self.mask_net = None
        # name -> dict(shape, init=(kind, kwargs)); order of creation kept (detector.py:57-65 semantics)
        self.params = []
        self.param_specs = {}
        self.weights = []
        self.biases = []
        self.do_not_update_params = []
        self.losses = []
        self.metrics = []
        self.roi_data_loader = None

    # ---- parameter bookkeeping -------------------------------------------------------------------
    def _param(self, name, shape, init, kind):
        if name not in self.param_specs:
            self.param_specs[name] = dict(shape=tuple(int(s) for s in shape), init=init)
            self.params.append(name)
            (self.weights if kind == 'w' else self.biases).append(name)
        else:
            assert self.param_specs[name]['shape'] == tuple(int(s) for s in shape), name
        return name

    def TrainableParams(self, gpu_id=-1):
        """Params that receive gradients: everything except AffineChannel scale/bias (gradient op has no
        dscale/dbias, affine_channel_nd_op.cc:29-37) and do_not_update params (detector.py:57-65)."""
        frozen = set(self.do_not_update_params)
        return [p for p in self.params if p not in frozen and not self.param_specs[p].get('affine')]

    # ---- convolutions --------------------------------------------------------------------------------
    def ConvNd(self, blob_in, blob_out, dim_in, dim_out, kernels, strides=None, pads=None, no_bias=0,
               weight_init=None, bias_init=None, group=1, dilations=1, weight=None, bias=None, **unused):
        assert group == 1, 'grouped conv not on the hot path (RESNETS.NUM_GROUPS == 1 in every config)'
        if not isinstance(dilations, int):
            assert all(d == 1 for d in dilations), 'dilated conv unsupported (MODEL.DILATION == 1 in every config)'
        else:
            assert dilations == 1
        kernels = [int(k) for k in kernels]
        strides = [1, 1, 1] if strides is None else [int(s) for s in strides]
        pads = [0] * 6 if pads is None else [int(p) for p in pads]
        assert pads[:3] == pads[3:], 'symmetric pads expected'
        assert strides[0] == 1, 'temporal stride unsupported (VIDEO.TIME_STRIDE_ON, FPN3D.py:199-203)'
        blob_in, blob_out = str(blob_in), str(blob_out)
        w = weight or self._param(blob_out + '_w', [dim_out, dim_in] + kernels,
                                  weight_init or ('XavierFill', {}), 'w')
        b = None
        if not no_bias:
            b = bias or self._param(blob_out + '_b', [dim_out], bias_init or ('ConstantFill', {'value': 0.}), 'b')
        return self.net.add(Op('Conv', [blob_in], [blob_out], w=w, b=b, scale=None, shift=None, dim_in=dim_in,
                               dim_out=dim_out, kernels=kernels, strides=strides[1:], pads=pads[:3], relu=False,
                               residual=None, res_mode=0))

    def Conv(self, blob_in, blob_out, dim_in, dim_out, kernel, stride=1, pad=0, no_bias=0, weight_init=None,
             bias_init=None, group=1, dilation=1, **kw):
        """2D conv (FPN.py:222-262, keypoint_rcnn_heads.py:61-66): recorded as a kT = 1 ConvNd; the parameter
        keeps the reference's 4-D shape [out, in, k, k]."""
        w = kw.get('weight') or self._param(str(blob_out) + '_w', [dim_out, dim_in, kernel, kernel],
                                             weight_init or ('XavierFill', {}), 'w')
        return self.ConvNd(blob_in, blob_out, dim_in, dim_out, [1, kernel, kernel], strides=[1, stride, stride],
                           pads=2 * [0, pad, pad], no_bias=no_bias, bias_init=bias_init, group=group,
                           dilations=dilation, weight=w, bias=kw.get('bias'))

    def ConvShared(self, blob_in, blob_out, dim_in, dim_out, kernel, weight=None, bias=None, nd=False, **kwargs):
        """detector.py:312-346: conv reusing another layer's parameters."""
        if nd:
            return self.ConvNd(blob_in, blob_out, dim_in, dim_out, kernel, weight=weight, bias=bias, **kwargs)
        return self.Conv(blob_in, blob_out, dim_in, dim_out, kernel, weight=weight, bias=bias, **kwargs)

    def AffineChannelNd(self, blob_in, blob_out, dim_out, share_with=None, inplace=False):
        """detector.py:89-108.  Folded into the producing conv's epilogue when possible."""
        if cfg.MODEL.USE_BN:
            raise NotImplementedError('SpatialBN path (MODEL.USE_BN) is not used by any shipped config')
        blob_in = str(blob_in)
        prefix = str(blob_out) if share_with is None else share_with
        s = self._param(prefix + '_s', [dim_out], ('ConstantFill', {'value': 1.}), 'w')
        b = self._param(prefix + '_b', [dim_out], ('ConstantFill', {'value': 0.}), 'b')
        self.param_specs[s]['affine'] = self.param_specs[b]['affine'] = True
        out = blob_in if inplace else str(blob_out)
        prod = self.net.producer(blob_in)
        if prod is not None and prod.type == 'Conv' and prod.args['scale'] is None and prod.args['b'] is None \
                and not prod.args['relu'] and not self.net.consumers(blob_in, prod):
            prod.args['scale'], prod.args['shift'] = s, b
            prod.outputs = [out]
            return out
        return self.net.add(Op('AffineChannel', [blob_in], [out], scale=s, shift=b))

    AffineChannel = AffineChannelNd

    def ConvAffineNd(self, blob_in, prefix, dim_in, dim_out, kernels, strides, pads, group=1, dilations=1,
                     weight_init=None, bias_init=None, suffix='_bn', inplace=False):
        """detector.py:410-438."""
        c = self.ConvNd(blob_in, prefix, dim_in, dim_out, kernels, strides=strides, pads=pads, group=group,
                        dilations=dilations, weight_init=weight_init, bias_init=bias_init, no_bias=1)
        return self.AffineChannelNd(c, prefix + suffix, dim_out, inplace=inplace)

    def ConvAffine(self, blob_in, prefix, dim_in, dim_out, kernel, stride, pad, group=1, dilation=1,
                   weight_init=None, bias_init=None, suffix='_bn', inplace=False):
        """detector.py:382-408."""
        c = self.Conv(blob_in, prefix, dim_in, dim_out, kernel, stride=stride, pad=pad, group=group,
                      dilation=dilation, weight_init=weight_init, bias_init=bias_init, no_bias=1)
        return self.AffineChannel(c, prefix + suffix, dim_out, inplace=inplace)

    def Relu(self, blob_in, blob_out):
        blob_in, blob_out = str(blob_in), str(blob_out)
        prod = self.net.producer(blob_in)
        if prod is not None and prod.type in ('Conv', 'FC') and not prod.args['relu'] \
                and not self.net.consumers(blob_in, prod):
            prod.args['relu'] = True
            prod.outputs = [blob_out]
            return blob_out
        return self.net.add(Op('Relu', [blob_in], [blob_out]))

    def StopGradient(self, blob_in, blob_out):
        """ResNet3D.py:273-274: no gradient flows below this blob (frozen conv1 / res2).  A marker: no kernel."""
        self.net.add(Op('StopGradient', [str(blob_in)], [str(blob_out)]))
        return str(blob_out)

    def MaxPool(self, blob_in, blob_out, kernels=None, pads=None, strides=None, kernel=None, pad=0, stride=1):
        if kernels is None:
            kernels, pads, strides = [1, kernel, kernel], 2 * [0, pad, pad], [1, stride, stride]
        assert kernels[0] == 1 and strides[0] == 1 and kernels[1] == kernels[2] and strides[1] == strides[2]
        return self.net.add(Op('MaxPool', [str(blob_in)], [str(blob_out)], k=int(kernels[1]), stride=int(strides[1]),
                               pad=int(pads[1])))

    def FC(self, blob_in, blob_out, dim_in, dim_out, weight_init=None, bias_init=None):
        blob_out = str(blob_out)
        w = self._param(blob_out + '_w', [dim_out, dim_in], weight_init or ('XavierFill', {}), 'w')
        b = self._param(blob_out + '_b', [dim_out], bias_init or ('ConstantFill', {'value': 0.}), 'b')
        return self.net.add(Op('FC', [str(blob_in)], [blob_out], w=w, b=b, dim_in=dim_in, dim_out=dim_out, relu=False))

    def Softmax(self, blob_in, blob_out, **kw):
        return self.net.add(Op('Softmax', [str(blob_in)], [str(blob_out)]))

    def ConvTranspose(self, blob_in, blob_out, dim_in, dim_out, kernel, pad=0, stride=1, group=1, weight_init=None,
                      bias_init=None, **kw):
        assert kernel == 4 and stride == 2 and pad == 1, 'only the k4/s2/p1 keypoint deconv is on the hot path (model_builder.py:848-856)'
        assert group >= 1 and dim_in % group == 0 and dim_out % group == 0, (dim_in, dim_out, group)
        blob_out = str(blob_out)
        # filter layout: Caffe2's (C_in, C_out / group, kH, kW); with cfg.HIP.DECONV_GROUP_IGNORED the full [dim_in, dim_out, k, k] blob
        # that brew.conv_transpose creates whatever the group (see workspace.Executor.op_ConvTranspose)
        per_group = dim_out // group if (group > 1 and not cfg.HIP.get('DECONV_GROUP_IGNORED', False)) else dim_out
        w = self._param(blob_out + '_w', [dim_in, per_group, kernel, kernel], weight_init or ('XavierFill', {}), 'w')
        b = self._param(blob_out + '_b', [dim_out], bias_init or ('ConstantFill', {'value': 0.}), 'b')
        extra = {'group': int(group)} if group > 1 else {}
        return self.net.add(Op('ConvTranspose', [str(blob_in)], [blob_out], w=w, b=b, dim_in=dim_in, dim_out=dim_out, **extra))

    def BilinearInterpolation(self, blob_in, blob_out, dim_in, dim_out, up_scale):
        """detector.py:348-380: fixed (non-trainable) bilinear ConvTranspose, kernel 2*up, stride up, pad up/2."""
        assert dim_in == dim_out
        assert up_scale % 2 == 0, 'Scale should be even'
        blob_out = str(blob_out)
        w = self._param(blob_out + '_w', [dim_in, dim_out, 2 * up_scale, 2 * up_scale],
                        ('BilinearFill', {'up_scale': up_scale}), 'w')
        b = self._param(blob_out + '_b', [dim_out], ('ConstantFill', {'value': 0.}), 'b')
        self.do_not_update_params += [w, b]
        return self.net.add(Op('BilinearInterpolation', [str(blob_in)], [blob_out], up_scale=int(up_scale), dim=dim_out))

    # ---- time <-> batch/channel helpers (detector.py:467-576) ---------------------------------------------
    # In NDHWC with frame-major storage these are metadata changes; they are recorded so that the wiring reads
    # like the reference and so that FetchBlob can return the reference layout.
    def MoveTimeToBatchDim(self, blob_in, blob_out=None):
        blob_out = blob_out or str(blob_in) + '_MovedTimeToBatchDim'
        return self.net.add(Op('TimeToBatch', [str(blob_in)], [str(blob_out)]))

    def GetTemporalDim(self, blob_in):
        """detector.py:440-465 makes a shape blob; here the frame count travels with the blob (workspace.Blob.T): a symbolic handle."""
        return ('T', str(blob_in))

    def MoveTimeToBatchDimInverse(self, blob_in, blob_out, temporal_dim):
        blob_out = blob_out or str(blob_in) + '_MovedTimeToBatchDimInv'
        return self.net.add(Op('BatchToTime', [str(blob_in)], [str(blob_out)], T=temporal_dim))

    def MoveTimeToChannelDim(self, blob_in, blob_out=None):
        blob_out = blob_out or str(blob_in) + '_MovedTimeToChDim'
        return self.net.add(Op('TimeToChannel', [str(blob_in)], [str(blob_out)]))

    def TimePool(self, blob_in, blob_out, pool_type):
        if pool_type != 'avg':
            raise NotImplementedError('Unknown type {}'.format(pool_type))
        blob_out = blob_out or str(blob_in) + '_TimePooled_avg'
        return self.net.add(Op('TimePoolAvg', [str(blob_in)], [str(blob_out)]))

    def SliceKeyFrame(self, blob_in, N):
        return self.net.add(Op('SliceKeyFrame', [str(blob_in)], [str(blob_in) + '_slicekey'], keyframe=int(N / 2)))

    def SpatialMean(self, blob_in, blob_out):
        """ReduceBackMean over W then H (+ExpandDims), ResNet3D.py:318-325."""
        return self.net.add(Op('SpatialMean', [str(blob_in)], [str(blob_out)]))

    def TimeMean(self, blob_in, blob_out):
        """The trailing ReduceBackMean over T of model_builder.py:439-440 / the TimePool of :532."""
        return self.net.add(Op('TimeMean', [str(blob_in)], [str(blob_out)]))

    # ---- Python-op wrappers ----------------------------------------------------------------------------------
    def GenerateProposals(self, blobs_in, blobs_out, anchors, spatial_scale):
        """detector.py:144-152.  blobs_in = [probs|logits, bbox_pred, im_info]."""
        return self.net.add(Op('GenerateProposals', [str(b) for b in blobs_in], [str(b) for b in blobs_out],
                               anchors=np.asarray(anchors), spatial_scale=float(spatial_scale)))

    def CollectAndDistributeFpnRpnProposals(self):
        """detector.py:168-204 (inference form)."""
        k_max, k_min = cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.RPN_MIN_LEVEL
        ins = ['rpn_rois_fpn%d' % l for l in range(k_min, k_max + 1)] + \
              ['rpn_roi_probs_fpn%d' % l for l in range(k_min, k_max + 1)]
        outs = ['rois'] + ['rois_fpn%d' % l for l in range(cfg.FPN.ROI_MIN_LEVEL, cfg.FPN.ROI_MAX_LEVEL + 1)] + \
               ['rois_idx_restore_int32']
        if self.train:   # detector.py:185-196: also reads roidb + im_info and emits the sampled training blobs
            ins += ['roidb', 'im_info']
            outs = ['rois', 'labels_int32', 'bbox_targets', 'bbox_inside_weights', 'bbox_outside_weights']
            if cfg.MODEL.KEYPOINTS_ON:
                outs += ['keypoint_rois', 'keypoint_locations_int32', 'keypoint_weights', 'keypoint_loss_normalizer']
        self.net.add(Op('CollectAndDistributeFpnRpnProposals', ins, outs, train=bool(self.train)))
        return outs

    def RoIFeatureTransform(self, blobs_in, blob_out, blob_rois='rois', method='RoIPoolF', resolution=7,
                            spatial_scale=1. / 16., sampling_ratio=0):
        """detector.py:256-310.  One op for single- and multi-level inputs, 2D and tube rois."""
        assert method == 'RoIAlign', 'only RoIAlign is used by the shipped configs (got {})'.format(method)
        multi = isinstance(blobs_in, list)
        ins = [str(b) for b in blobs_in] if multi else [str(blobs_in)]
        scales = list(spatial_scale) if multi else [spatial_scale]
        if multi:
            # reference order is coarse -> fine (blobs_in[k_max - lvl]); store fine -> coarse
            ins, scales = ins[::-1], scales[::-1]
        return self.net.add(Op('RoIFeatureTransform', ins + [str(blob_rois)], [str(blob_out)], n_feat=len(ins),
                               scales=[float(s) for s in scales], resolution=int(resolution),
                               sampling_ratio=int(sampling_ratio)))

    # ---- misc parity with the reference helper -----------------------------------------------------------------
    def DropoutIfTraining(self, blob_in):
        return blob_in

    def UpdateWorkspaceLr(self, cur_iter):
        from detectandtrack_amd.utils import lr_policy
        return lr_policy.get_lr_at_iter(cur_iter)
