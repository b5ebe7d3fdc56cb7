"""Oracle: the detector forward graph in torch-CPU fp32 (test infrastructure).

PARITY UNPINNED for the arithmetic (Caffe2 b4e1588 ops, see oracle/__init__.py);
the WIRING follows the reference builders line by line:

  body        lib/modeling/ResNet3D.py:21-298 (bottleneck :21-55, basic :59-82,
              shortcut :89-101, block :120-154, stage :210-227, body :251-298)
  FPN         lib/modeling/FPN3D.py:109-222
  body->head  lib/modeling/model_builder.py:1024-1042 (time_pool_blobs),
              lib/modeling/detector.py:559-576 (TimePool / SliceKeyFrame)
  FPN RPN     lib/modeling/FPN.py:205-279
  proposals   oracle.proposals (generate_proposals.py, collect_and_distribute...)
  box head    lib/modeling/head_builder.py:17-38, model_builder.py:426-478
  C4 RPN      lib/modeling/model_builder.py:500-609 (tube path, nd=True; 2D path, nd=False)
  C4 box head lib/modeling/ResNet3D.py:301-327, model_builder.py:427-473
  kps head    lib/modeling/keypoint_rcnn_heads.py:39-73,
              model_builder.py:755-870, detector.py:348-380

All blobs are NC[T]HW fp32 exactly like the reference workspace; parameter names
are the reference's blob names (`conv1_w`, `res2_0_branch2a_bn_s`, ...).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import proposals as prop
from .anchors import generate_anchors
from .roi_align import roi_align_2d, roi_align_tube


def _t(a):
    if isinstance(a, torch.Tensor):   # oracle/train_ref.py passes leaf tensors so that autograd sees the weights
        return a
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class Net(object):
    """Holds weights (dict name -> ndarray) and options; runs the forward."""

    def __init__(self, weights, opts):
        self.w = weights
        self.o = opts
        self.blobs = {}

    # ---- primitive ops (public Caffe2 semantics) -------------------------
    def conv_nd(self, x, name, kernels, strides, pads, bias=True):
        # Caffe2 ConvNd == cross-correlation, symmetric explicit pads
        w = _t(self.w[name + '_w'])
        b = _t(self.w[name + '_b']) if bias else None
        assert list(w.shape[2:]) == list(kernels), (name, w.shape, kernels)
        return F.conv3d(x, w, b, stride=tuple(strides), padding=tuple(pads))

    def conv2d(self, x, wname, k, stride=1, pad=0, bname=None):
        w = _t(self.w[wname + '_w'])
        b = _t(self.w[(bname or wname) + '_b'])
        return F.conv2d(x, w, b, stride=stride, padding=pad)

    def affine(self, x, name):
        # ops/affine_channel_nd_op.cu:20-32: out = in*scale[c] + bias[c]
        s = _t(self.w[name + '_s'])
        b = _t(self.w[name + '_b'])
        shp = [1, -1] + [1] * (x.dim() - 2)
        return x * s.view(shp) + b.view(shp)

    def conv_affine_nd(self, x, prefix, kernels, strides, pads):
        # detector.py:410-438 (ConvNd no_bias + AffineChannelNd '<prefix>_bn')
        return self.affine(self.conv_nd(x, prefix, kernels, strides, pads, bias=False), prefix + '_bn')

    def fc(self, x, name):
        w = _t(self.w[name + '_w'])
        b = _t(self.w[name + '_b'])
        return F.linear(x.reshape(x.shape[0], -1), w, b)

    # ---- ResNet3D body -----------------------------------------------------
    def _basic(self, x, prefix, stride, kt):
        # ResNet3D.py:59-82
        y = self.conv_affine_nd(x, prefix + '_branch2a', [kt, 3, 3], [1, stride, stride], [kt // 2, 1, 1])
        y = F.relu(y)
        return self.conv_affine_nd(y, prefix + '_branch2b', [kt, 3, 3], [1, 1, 1], [kt // 2, 1, 1])

    def _bottleneck(self, x, prefix, stride, kt):
        # ResNet3D.py:21-55, STRIDE_1X1 True (config.py RESNETS.STRIDE_1X1)
        y = F.relu(self.conv_affine_nd(x, prefix + '_branch2a', [1, 1, 1], [1, stride, stride], [0, 0, 0]))
        y = F.relu(self.conv_affine_nd(y, prefix + '_branch2b', [kt, 3, 3], [1, 1, 1], [kt // 2, 1, 1]))
        return self.conv_affine_nd(y, prefix + '_branch2c', [1, 1, 1], [1, 1, 1], [0, 0, 0])

    def _stage(self, x, stage_id, prefix, n, dim_in, dim_out, kt, stride_init=2):
        # ResNet3D.py:120-154, 210-227
        trans = self._basic if self.o['trans'] == 'basic' else self._bottleneck
        for i in range(n):
            p = '{}_{}'.format(prefix, i)
            stride = stride_init if (dim_in != dim_out and stage_id != 1) else 1
            tr = trans(x, p, stride, kt)
            if dim_in == dim_out:
                sc = x
            else:  # ResNet3D.py:89-101
                sc = self.affine(self.conv_nd(x, p + '_branch1', [1, 1, 1], [1, stride, stride],
                                              [0, 0, 0], bias=False), p + '_branch1_bn')
            x = F.relu(tr + sc)
            self.blobs[p + '_sum'] = x
            dim_in = dim_out
        return x

    def body(self, data):
        # ResNet3D.py:251-298
        o = self.o
        dims = o['feat_dims']
        kt = o['kt_body']
        x = self.conv_nd(data, 'conv1', [1, 7, 7], [1, 2, 2], [0, 3, 3], bias=False)
        x = F.relu(self.affine(x, 'res_conv1_bn'))
        x = F.max_pool3d(x, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self.blobs['pool1'] = x
        counts = o['block_counts']
        x = self._stage(x, 1, 'res2', counts[0], dims[0], dims[1], 1)
        x = self._stage(x, 2, 'res3', counts[1], dims[1], dims[2], kt)
        x = self._stage(x, 3, 'res4', counts[2], dims[2], dims[3], kt)
        if len(counts) == 4:
            x = self._stage(x, 4, 'res5', counts[3], dims[3], dims[4], kt)
        return x

    # ---- FPN3D -----------------------------------------------------------------
    def fpn(self):
        # FPN3D.py:109-183; stage blobs coarse -> fine
        o = self.o
        kt = o['kt_body']
        counts = o['block_counts']
        names = ['res{}_{}_sum'.format(s + 2, counts[s] - 1) for s in (3, 2, 1, 0)]
        inner = {}
        inner[names[0]] = self.conv_nd(self.blobs[names[0]], 'fpn_inner_' + names[0], [1, 1, 1], [1, 1, 1], [0, 0, 0])
        for i in range(3):
            top = inner[names[i]]
            lat = self.conv_nd(self.blobs[names[i + 1]], 'fpn_inner_' + names[i + 1] + '_lateral',
                               [1, 1, 1], [1, 1, 1], [0, 0, 0])
            # FPN3D.py:207-222: time->channel, nearest 2x in H,W, channel->time, Sum
            td = top.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
            inner[names[i + 1]] = lat + td
        out = []
        for i in range(4):
            out.append(self.conv_nd(inner[names[i]], 'fpn_' + names[i], [kt, 3, 3], [1, 1, 1], [kt // 2, 1, 1]))
            self.blobs['fpn_' + names[i]] = out[-1]
        # P6: FPN3D.py:155-164 MaxPool k=1 s=[1,2,2]
        p6 = out[0][:, :, :, ::2, ::2]
        self.blobs['fpn_' + names[0] + '_subsampled_2x'] = p6
        return [p6] + out  # [P6, P5, P4, P3, P2]

    def time_link(self, blobs):
        # model_builder.py:1024-1042; detector.py:559-576
        link = self.o['body_head_link']
        if link == '':
            return blobs
        res = []
        for b in blobs:
            if link == 'avg':
                res.append(b.mean(dim=2))
            elif link == 'slice-center':
                k = int(self.o['num_frames_mid'] / 2)
                res.append(b[:, :, k])
            else:
                raise NotImplementedError(link)
        return res

    # ---- FPN RPN (2D heads) ------------------------------------------------------
    def fpn_rpn(self, blobs2d, im_info):
        # FPN.py:205-279; blobs2d ordered [P6..P2]; weights shared from level k_min=2
        o = self.o
        k_min, k_max = 2, 6
        rois_l, probs_l = [], []
        for lvl in range(k_min, k_max + 1):
            x = blobs2d[k_max - lvl]
            h = F.relu(self.conv2d(x, 'conv_rpn_fpn2', 3, 1, 1))
            logits = self.conv2d(h, 'rpn_cls_logits_fpn2', 1)
            deltas = self.conv2d(h, 'rpn_bbox_pred_fpn2', 1)
            probs = torch.sigmoid(logits)
            self.blobs['rpn_cls_probs_fpn%d' % lvl] = probs
            self.blobs['rpn_bbox_pred_fpn%d' % lvl] = deltas
            anchors = generate_anchors(stride=2. ** lvl, sizes=(o['rpn_anchor_start'] * 2. ** (lvl - k_min),),
                                       aspect_ratios=o['rpn_aspect_ratios'], time_dim=1)
            r, p = prop.generate_proposals(probs.numpy(), deltas.numpy(), im_info, anchors, 1. / 2. ** lvl,
                                           o['pre_nms_topn'], o['post_nms_topn'], o['rpn_nms_thresh'],
                                           o['rpn_min_size'])
            rois_l.append(r)
            probs_l.append(p)
        rois = prop.collect(rois_l, probs_l, o['post_nms_topn'])
        return prop.distribute(rois, 2, 5)

    def roi_feat_fpn(self, blobs2d_p5_to_p2, per_level_rois, restore, pooled, sampling):
        # detector.py:256-310 (multi-level branch), blobs ordered [P5, P4, P3, P2]
        outs = []
        for lvl in range(2, 6):
            f = blobs2d_p5_to_p2[5 - lvl].numpy()
            outs.append(roi_align_2d(f, per_level_rois[lvl - 2], pooled, 1. / 2. ** lvl, sampling))
        cat = np.concatenate(outs, axis=0)
        return cat[restore]  # BatchPermutation Y = X[I]

    def box_head_2mlp(self, roi_feat):
        # head_builder.py:17-38 + model_builder.py:426-478 (2D branch)
        x = _t(roi_feat)
        x = F.relu(self.fc(x, 'fc6'))
        x = F.relu(self.fc(x, 'fc7'))
        cls_prob = F.softmax(self.fc(x, 'cls_score'), dim=1)
        bbox_pred = self.fc(x, 'bbox_pred')
        return cls_prob.numpy(), bbox_pred.numpy()

    def kps_head_2d(self, roi_feat):
        # keypoint_rcnn_heads.py:39-69 (nd=False) + model_builder.py:755-870
        x = _t(roi_feat)
        for i in range(self.o['kps_num_convs']):
            x = F.relu(self.conv2d(x, 'conv_fcn%d' % (i + 1), 3, 1, 1))
        return self.kps_outputs_2d(x)

    def kps_outputs_2d(self, x):
        # model_builder.py:848-868: ConvTranspose k4 s2 p1 -> kps_score_lowres;
        # detector.py:348-380: fixed bilinear ConvTranspose k=2*up, s=up, p=up/2
        w = _t(self.w['kps_score_lowres_w'])
        b = _t(self.w['kps_score_lowres_b'])
        low = F.conv_transpose2d(x, w, b, stride=2, padding=1)
        self.blobs['kps_score_lowres'] = low
        up = self.o['kps_up_scale']
        K = low.shape[1]
        return F.conv_transpose2d(low, _t(bilinear_kernel(K, up)), None, stride=up, padding=up // 2)

    # ---- C4 path of the 2D models (configs/video/3d/01_R-18_*.yaml, 02_R-18_*.yaml) ------------------------------
    def rpn_c4_2d(self, feat2d, im_info):
        # model_builder.py:500-609 with nd=False: Conv 3x3 + Relu, 1x1 logits (A) and deltas (4A), Sigmoid, GenerateProposals
        o = self.o
        anchors = generate_anchors(stride=16., sizes=o['rpn_sizes'], aspect_ratios=o['rpn_c4_aspect_ratios'], time_dim=1)
        h = F.relu(self.conv2d(feat2d, 'conv_rpn', 3, 1, 1))
        probs = torch.sigmoid(self.conv2d(h, 'rpn_cls_logits', 1))
        deltas = self.conv2d(h, 'rpn_bbox_pred', 1)
        self.blobs['rpn_cls_probs'] = probs
        self.blobs['rpn_bbox_pred'] = deltas
        return prop.generate_proposals(probs.numpy(), deltas.numpy(), im_info, anchors, 1. / 16., o['pre_nms_topn'],
                                       o['post_nms_topn'], o['rpn_nms_thresh'], o['rpn_min_size'])

    def box_head_c4_2d(self, feat2d, rois):
        # ResNet.py:268-287 (RoIAlign, res5 per RoI, AveragePool 7) + model_builder.py:426-478 (2D branch: FC scores / deltas)
        o = self.o
        pooled = o['frcn_res']
        x = _t(roi_align_2d(feat2d.numpy(), rois, pooled, 1. / 16., o['frcn_sampling']))[:, :, None]
        dims = o['feat_dims']
        x = self._stage(x, 4, 'res5', o['res5_blocks'], dims[3], o['res5_dim'], 1, stride_init=int(pooled / 7))
        x = x[:, :, 0].mean(dim=3).mean(dim=2)
        self.blobs['res5_pool'] = x
        return F.softmax(self.fc(x, 'cls_score'), dim=1).numpy(), self.fc(x, 'bbox_pred').numpy()

    def kps_head_c4_2d(self, feat2d, kp_rois):
        # keypoint_rcnn_heads.py:39-69 (nd=False) on the C4 feature map
        o = self.o
        return self.kps_head_2d(roi_align_2d(feat2d.numpy(), kp_rois, o['kps_res'], 1. / 16., o['kps_sampling']))

    # ---- C4 tube path ----------------------------------------------------------------
    def rpn_c4_tube(self, feat, im_info):
        # model_builder.py:500-609 with nd=True
        o = self.o
        T = o['num_frames_mid']
        ktr = o['kt_rpn']
        anchors = generate_anchors(stride=16., sizes=o['rpn_sizes'], aspect_ratios=o['rpn_c4_aspect_ratios'],
                                   time_dim=T)
        A = anchors.shape[0]
        h = F.relu(self.conv_nd(feat, 'conv_rpn', [ktr, 3, 3], [1, 1, 1], [ktr // 2, 1, 1]))
        logits = self.conv_nd(h, 'rpn_cls_logits_1', [1, 1, 1], [1, 1, 1], [0, 0, 0]).mean(dim=2)
        d = self.conv_nd(h, 'rpn_bbox_pred_1', [1, 1, 1], [1, 1, 1], [0, 0, 0])  # N,(A*4),T,H,W
        N, _, Tt, H, W = d.shape
        # model_builder.py:552-563: N,(A,4),T,H,W -> N,A,T,4,H,W -> N,(A*T*4),H,W
        d = d.reshape(N, A, 4, Tt, H, W).permute(0, 1, 3, 2, 4, 5).reshape(N, A * Tt * 4, H, W)
        probs = torch.sigmoid(logits)
        self.blobs['rpn_cls_probs'] = probs
        self.blobs['rpn_bbox_pred'] = d
        return prop.generate_proposals(probs.numpy(), d.numpy(), im_info, anchors, 1. / 16.,
                                       o['pre_nms_topn'], o['post_nms_topn'], o['rpn_nms_thresh'],
                                       o['rpn_min_size'])

    def box_head_c4_tube(self, feat, rois):
        # ResNet3D.py:301-327 + model_builder.py:427-473 (is_head_3d)
        o = self.o
        pooled = o['frcn_res']
        x = _t(roi_align_tube(feat.numpy(), rois, pooled, 1. / 16., o['frcn_sampling']))
        dims = o['feat_dims']
        x = self._stage(x, 4, 'res5', o['res5_blocks'], dims[3], o['res5_dim'], 1, stride_init=int(pooled / 7))
        x = x.mean(dim=4).mean(dim=3)[:, :, :, None, None]  # ReduceBackMean x2 + ExpandDims
        cls = self.conv_nd(x, 'cls_score_1', [1, 1, 1], [1, 1, 1], [0, 0, 0])
        cls = cls.mean(dim=4).mean(dim=3).mean(dim=2)
        cls_prob = F.softmax(cls, dim=1)
        bp = self.conv_nd(x, 'bbox_pred_1', [1, 1, 1], [1, 1, 1], [0, 0, 0])  # R,(K*4),T,1,1
        R, K4, T, H, W = bp.shape
        bp = bp.reshape(R, K4 // 4, 4, T, H, W).permute(0, 1, 3, 2, 4, 5).reshape(R, -1, H, W)
        bp = bp.mean(dim=3).mean(dim=2)
        return cls_prob.numpy(), bp.numpy()

    def kps_head_tube(self, feat, kp_rois):
        # keypoint_rcnn_heads.py:39-73 nd=True; model_builder.py:755-870 with
        # NO_3D_DECONV_TIME_TO_CH (time->batch, 2D deconvs, batch->time, time->channel)
        o = self.o
        kt = o['kt_kps']
        x = _t(roi_align_tube(feat.numpy(), kp_rois, o['kps_res'], 1. / 16., o['kps_sampling']))
        for i in range(o['kps_num_convs']):
            x = F.relu(self.conv_nd(x, 'conv_fcn%d' % (i + 1), [kt, 3, 3], [1, 1, 1], [kt // 2, 1, 1]))
        return self.kps_outputs_tube(x)

    def kps_outputs_tube(self, x):
        """model_builder.py:755-870 on a 3D head output (R, C, T, H, W).  o['kps_time_to_ch'] False: KRCNN.NO_3D_DECONV_TIME_TO_CH True
        (:760-764 time -> batch, the 2D deconvs with shared weights, :864-868 batch -> time, time -> channel).  True: the reference default
        (:765-767 MoveTimeToChannelDim = detector.py:480-491 Transpose(0,2,1,3,4) + Reshape: channel t*C + c; :848-856 ConvTranspose
        dim*T -> K*T with group = T; :858-863 the bilinear deconv over K*T maps).  The filter decides what `group` means: Caffe2's grouped
        layout (T*C, K, 4, 4) -> groups = T; the full (T*C, T*K, 4, 4) blob brew.conv_transpose creates -> the argument dropped (dense), which
        is what a Caffe2 without ConvTranspose group support executes."""
        R, C, T, H, W = x.shape
        if not self.o.get('kps_time_to_ch', False):
            xb = x.permute(0, 2, 1, 3, 4).reshape(R * T, C, H, W)
            y = self.kps_outputs_2d(xb)  # (R*T, 17, 56, 56)
            K, M = y.shape[1], y.shape[2]
            # batch->time: (R, T, K, M, M) -> (R, K, T, M, M); time->channel: -> (R, T*K, M, M)
            return y.reshape(R, T, K, M, M).reshape(R, T * K, M, M)
        x2 = x.permute(0, 2, 1, 3, 4).reshape(R, T * C, H, W)
        w = _t(self.w['kps_score_lowres_w'])
        b = _t(self.w['kps_score_lowres_b'])
        assert w.shape[0] == T * C and b.shape[0] % T == 0, (w.shape, b.shape, T, C)
        groups = T if w.shape[1] * T == b.shape[0] else 1
        assert w.shape[1] * groups == b.shape[0]
        low = F.conv_transpose2d(x2, w, b, stride=2, padding=1, groups=groups)
        self.blobs['kps_score_lowres'] = low
        up = self.o['kps_up_scale']
        return F.conv_transpose2d(low, _t(bilinear_kernel(low.shape[1], up)), None, stride=up, padding=up // 2)


    # ---- FPN tube path (declared extension) -----------------------------------------------
    def fpn_rpn_tube(self, pyr, im_info):
        # the design of the reference's dead FPN3D.py:232-330: kT x 3 x 3 conv + ReLU per level, time -> channels
        # (detector.py:480-491: channel t*C + c), 2D 1x1 heads over C*T inputs; weights shared from level 2
        o = self.o
        T, kt = o['num_frames_mid'], o['kt_rpn']
        k_min, k_max = 2, 6
        rois_l, probs_l = [], []
        for lvl in range(k_min, k_max + 1):
            x = pyr[k_max - lvl]
            w, b = _t(self.w['conv_rpn_fpn2_w']), _t(self.w['conv_rpn_fpn2_b'])
            h = F.relu(F.conv3d(x, w, b, stride=1, padding=(kt // 2, 1, 1)))
            N, C, Tt, H, W = h.shape
            h2 = h.permute(0, 2, 1, 3, 4).reshape(N, Tt * C, H, W)
            logits = self.conv2d(h2, 'rpn_cls_logits_fpn2', 1)
            deltas = self.conv2d(h2, 'rpn_bbox_pred_fpn2', 1)
            probs = torch.sigmoid(logits)
            self.blobs['rpn_cls_probs_fpn%d' % lvl] = probs
            self.blobs['rpn_bbox_pred_fpn%d' % lvl] = deltas
            anchors = generate_anchors(stride=2. ** lvl, sizes=(o['rpn_anchor_start'] * 2. ** (lvl - k_min),),
                                       aspect_ratios=o['rpn_aspect_ratios'], time_dim=T)
            r, p = prop.generate_proposals(probs.numpy(), deltas.numpy(), im_info, anchors, 1. / 2. ** lvl,
                                           o['pre_nms_topn'], o['post_nms_topn'], o['rpn_nms_thresh'],
                                           o['rpn_min_size'])
            rois_l.append(r)
            probs_l.append(p)
        return prop.collect(rois_l, probs_l, o['post_nms_topn'])

    def roi_feat_fpn_tube(self, pyr_p5_to_p2, rois, pooled, sampling):
        # detector.py:256-310 on tube rois: level by mean area over the frames (FPN.py:349-360), RoIAlign per frame
        lvls = prop.map_rois_to_fpn_levels(rois[:, 1:], 2, 5)
        out = None
        for lvl in range(2, 6):
            idx = np.where(lvls == lvl)[0]
            if len(idx) == 0:
                continue
            f = roi_align_tube(pyr_p5_to_p2[5 - lvl].numpy(), rois[idx], pooled, 1. / 2. ** lvl, sampling)
            if out is None:
                out = np.zeros((rois.shape[0],) + f.shape[1:], dtype=np.float32)
            out[idx] = f
        return out  # (R, C, T, P, P)

    def box_head_2mlp_tube(self, roi_feat):
        # head_builder.py:17-38 with fc6 over T*C*res*res (c, t, h, w order); outputs: K scores, K*T*4 deltas
        x = _t(roi_feat).reshape(roi_feat.shape[0], -1)
        x = F.relu(self.fc(x, 'fc6'))
        x = F.relu(self.fc(x, 'fc7'))
        return F.softmax(self.fc(x, 'cls_score'), dim=1).numpy(), self.fc(x, 'bbox_pred').numpy()

    def kps_head_tube_feat(self, roi_feat):
        # keypoint_rcnn_heads.py:39-73 nd=True on (R, C, T, 14, 14) tube features; see kps_head_tube
        o = self.o
        kt = o['kt_kps']
        x = _t(roi_feat)
        for i in range(o['kps_num_convs']):
            x = F.relu(self.conv_nd(x, 'conv_fcn%d' % (i + 1), [kt, 3, 3], [1, 1, 1], [kt // 2, 1, 1]))
        return self.kps_outputs_tube(x)


def bilinear_kernel(dim, up_scale):
    """detector.py:356-372 (diagonal bilinear deconv kernel, size 2*up)."""
    size = up_scale * 2
    factor = (size + 1) // 2
    center = factor - 1 if size % 2 == 1 else factor - 0.5
    og = np.ogrid[:size, :size]
    filt = (1 - abs(og[0] - center) / factor) * (1 - abs(og[1] - center) / factor)
    k = np.zeros((dim, dim, size, size), dtype=np.float32)
    k[range(dim), range(dim), :, :] = filt
    return k


DEFAULT_OPTS = dict(
    trans='basic', block_counts=(2, 2, 2, 2), feat_dims=(64, 64, 128, 256, 512),
    kt_body=3, kt_rpn=3, kt_kps=3, body_head_link='slice-center', num_frames_mid=1,
    rpn_anchor_start=32, rpn_aspect_ratios=(0.5, 1, 2), pre_nms_topn=1000, post_nms_topn=1000,
    rpn_nms_thresh=0.7, rpn_min_size=0, rpn_sizes=(64, 128, 256, 512), rpn_c4_aspect_ratios=(0.5, 1, 2),
    frcn_res=7, frcn_sampling=2, kps_res=14, kps_sampling=2, kps_num_convs=8, kps_up_scale=2,
    res5_blocks=2, res5_dim=512, kps_time_to_ch=False,
)


def opts_for(arch, **kw):
    o = dict(DEFAULT_OPTS)
    if arch == 'R18':
        o.update(trans='basic', block_counts=(2, 2, 2, 2), feat_dims=(64, 64, 128, 256, 512),
                 res5_blocks=2, res5_dim=512)
    elif arch == 'R50':
        o.update(trans='bottleneck', block_counts=(3, 4, 6, 3), feat_dims=(64, 256, 512, 1024, 2048),
                 res5_blocks=3, res5_dim=2048)
    elif arch == 'R101':
        o.update(trans='bottleneck', block_counts=(3, 4, 23, 3), feat_dims=(64, 256, 512, 1024, 2048),
                 res5_blocks=3, res5_dim=2048)
    else:
        raise ValueError(arch)
    o.update(kw)
    return o
