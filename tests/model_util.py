"""Shared helpers for the end-to-end model tests / smoke(): build the product model + the oracle on the same
synthetic weights and inputs."""
import numpy as np


def fpn3d_kps_cfg(arch='18', T=4, kt=3, link='slice-center', pre=300, post=100, dtype='fp32'):
    return {
        'MODEL': {'TYPE': 'keypoint_rcnn', 'CONV_BODY': 'FPN3D.add_fpn_ResNet%s_conv5_body' % arch,
                  'ROI_HEAD': 'head_builder.add_roi_2mlp_head', 'NUM_CLASSES': 2, 'FASTER_RCNN': True,
                  'KEYPOINTS_ON': True, 'VIDEO_ON': True},
        'FPN': {'FPN_ON': True, 'MULTILEVEL_ROIS': True, 'MULTILEVEL_RPN': True},
        'FAST_RCNN': {'ROI_XFORM_METHOD': 'RoIAlign', 'ROI_XFORM_RESOLUTION': 7, 'ROI_XFORM_SAMPLING_RATIO': 2},
        'KRCNN': {'ROI_KEYPOINTS_HEAD': 'keypoint_rcnn_heads.add_roi_pose_head_v1convX', 'NUM_STACKED_CONVS': 8,
                  'NUM_KEYPOINTS': 17, 'USE_DECONV_OUTPUT': True, 'CONV_INIT': 'MSRAFill', 'CONV_HEAD_DIM': 512,
                  'UP_SCALE': 2, 'HEATMAP_SIZE': 56, 'ROI_XFORM_METHOD': 'RoIAlign', 'ROI_XFORM_RESOLUTION': 14,
                  'ROI_XFORM_SAMPLING_RATIO': 2},
        'VIDEO': {'NUM_FRAMES': T, 'TIME_KERNEL_DIM': kt, 'BODY_HEAD_LINK': link, 'WEIGHTS_INFLATE_MODE': 'center-only'},
        'TEST': {'RPN_PRE_NMS_TOP_N': pre, 'RPN_POST_NMS_TOP_N': post, 'COMPETITION_MODE': False, 'NMS': 0.5},
        'HIP': {'DTYPE': dtype},
    }


def build_product(cfg_dict, seed=3):
    """Returns (model, workspace, weights dict).  Needs the GPU."""
    from detectandtrack_amd.core.config import cfg, cfg_from_cfg, assert_and_infer_cfg, reset_cfg
    from detectandtrack_amd.modeling import model_builder
    from detectandtrack_amd.utils import net as net_utils
    from detectandtrack_amd import workspace
    reset_cfg()
    cfg_from_cfg(cfg_dict)
    assert_and_infer_cfg()
    model = model_builder.create(cfg.MODEL.TYPE, train=False)
    workspace.ResetWorkspace()
    ws = workspace.GlobalWorkspace()
    weights = net_utils.synthetic_params(model, seed)
    for k, v in weights.items():
        ws.set_param(k, v)
    ws.CreateNet(model.net)
    ws.CreateNet(model.conv_body_net)
    if model.keypoint_net is not None:
        ws.CreateNet(model.keypoint_net)
    return model, ws, weights


def synthetic_clip(T, H, W, seed=3):
    """uint8-like BGR frames minus PIXEL_MEANS, NC(T)HW fp32."""
    rs = np.random.RandomState(seed)
    means = np.array([102.9801, 115.9465, 122.7717], dtype=np.float32).reshape(1, 3, 1, 1, 1)
    # smooth-ish content so activations are not pure noise
    base = rs.uniform(0, 255, (1, 3, T, H // 8 + 1, W // 8 + 1)).astype(np.float32)
    data = np.repeat(np.repeat(base, 8, axis=3), 8, axis=4)[:, :, :, :H, :W]
    data = data + rs.uniform(-20, 20, data.shape).astype(np.float32)
    return np.clip(data, 0, 255) - means


def oracle_opts(arch, T, kt, link, pre, post):
    from oracle.net3d import opts_for
    return opts_for('R' + arch, kt_body=kt, body_head_link=link, num_frames_mid=T, pre_nms_topn=pre,
                    post_nms_topn=post)


def c4_tube_kps_cfg(T=3, kt=3, pre=300, post=60, dtype='fp32', deconv='time_to_batch'):
    """The shipped 3D configuration (configs/video/3d/04_R-18-3D_PTFromImNet.yaml): ResNet-18 3D C4 body, tube RPN,
    per-RoI res5 head, 3D keypoint head.  deconv: 'time_to_batch' (KRCNN.NO_3D_DECONV_TIME_TO_CH True, what the shipped configs
    set) | 'grouped' (the reference default: time -> channels + ConvTranspose group = T) | 'group_ignored' (the same graph with the
    full brew filter and the group argument dropped, cfg.HIP.DECONV_GROUP_IGNORED)."""
    d = fpn3d_kps_cfg('18', T=T, kt=kt, link='', pre=pre, post=post, dtype=dtype)
    d['MODEL'].update(CONV_BODY='ResNet3D.add_ResNet18_conv4_body', ROI_HEAD='ResNet3D.add_ResNet18_roi_conv5_head')
    d['FPN'] = {'FPN_ON': False, 'MULTILEVEL_ROIS': False, 'MULTILEVEL_RPN': False}
    assert deconv in ('time_to_batch', 'grouped', 'group_ignored')
    d['KRCNN'].update(ROI_KEYPOINTS_HEAD='keypoint_rcnn_heads.add_roi_pose_head_v1convX_3d',
                      NO_3D_DECONV_TIME_TO_CH=(deconv == 'time_to_batch'))
    d['HIP']['DECONV_GROUP_IGNORED'] = (deconv == 'group_ignored')
    return d


def fpn3d_tube_kps_cfg(T=2, kt=3, pre=200, post=50, dtype='fp32'):
    """Declared extension: FPN3D body kept 3D (BODY_HEAD_LINK ''), tube RPN per level, tube rois
    on the 2-MLP box head, 3D keypoint head."""
    d = fpn3d_kps_cfg('18', T=T, kt=kt, link='', pre=pre, post=post, dtype=dtype)
    d['KRCNN'].update(ROI_KEYPOINTS_HEAD='keypoint_rcnn_heads.add_roi_pose_head_v1convX_3d',
                      NO_3D_DECONV_TIME_TO_CH=True)
    return d


def fpn2d_kps_cfg(arch='50', pre=1000, post=1000, dtype='fp32'):
    """BASELINE configs 1-2: the pure 2D R-50-FPN keypoint R-CNN (MODEL.VIDEO_ON False; reference lib/modeling/FPN.py:114-202,
    ResNet.py:231-266; shipped as configs/video/2d_best/01_R101_best_hungarian.yaml with the R-101 body)."""
    d = fpn3d_kps_cfg(arch, T=1, kt=1, link='', pre=pre, post=post, dtype=dtype)
    d['MODEL'].update(CONV_BODY='FPN.add_fpn_ResNet%s_conv5_body' % arch, VIDEO_ON=False)
    d.pop('VIDEO')
    return d


def oracle_weights_2d(weights):
    """The oracle graph is written on 5-D blobs: 2D conv weights of the body / FPN ([o, i, k, k]) get a unit time axis."""
    out = {}
    for k, v in weights.items():
        v = np.asarray(v)
        if v.ndim == 4 and k.endswith('_w') and k.startswith(('conv1', 'res', 'fpn_')):
            v = v[:, :, None]
        out[k] = v
    return out


def check_c4_tube_against_oracle(model, ws, weights, T, H, W, pre, post, kps_time_to_ch=False, im_scale=1.0, n_kp=5, max_box_rois=None):
    """The 3D C4 tube model (ResNet-18 3D C4 body -> tube RPN -> tube RoIAlign -> per-RoI res5 -> 3D keypoint head) in fp32 parity mode
    against the oracle graph on one synthetic clip of T x H x W: res4 features, the fused RPN head (T-averaged logits, per-frame deltas),
    proposals (same count, device tubes found in the oracle set), box head on the device tubes, kps_score < 1e-3 on the first tubes."""
    import torch
    from oracle.net3d import Net, opts_for
    data = synthetic_clip(T, H, W)
    im_info = np.array([[H, W, im_scale]], dtype=np.float32)
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', im_info)
    ws.RunNet(model.net.name)
    net = Net(weights, opts_for('R18', block_counts=(2, 2, 2), kt_body=3, kt_rpn=3, kt_kps=3, body_head_link='',
                                num_frames_mid=T, pre_nms_topn=pre, post_nms_topn=post, kps_time_to_ch=kps_time_to_ch))
    feat = net.body(torch.from_numpy(data))
    got = ws.FetchBlob('res4_1_sum')
    assert got.shape == tuple(feat.shape)
    assert np.abs(got - feat.numpy()).max() < 1e-3 * max(1.0, float(feat.abs().max()))
    ref_rois, _ = net.rpn_c4_tube(feat, im_info)[:2]
    # fused head: A objectness logits + 4A deltas per frame
    head = ws.FetchBlob('rpn_cls_logits_1+rpn_bbox_pred_1')          # (1, 5A, T, h, w)
    A = 12
    assert head.shape[1] == 5 * A and head.shape[2] == T
    probs = 1.0 / (1.0 + np.exp(-head[:, :A].mean(axis=2)))
    np.testing.assert_allclose(probs, net.blobs['rpn_cls_probs'].numpy(), atol=1e-4)
    d = head[:, A:].reshape(1, A, 4, T, head.shape[3], head.shape[4]).transpose(0, 1, 3, 2, 4, 5)
    np.testing.assert_allclose(d.reshape(1, A * T * 4, head.shape[3], head.shape[4]),
                               net.blobs['rpn_bbox_pred'].numpy(), atol=1e-3)
    rois = ws.FetchBlob('rois')
    assert rois.shape[1] == 4 * T + 1 and rois.shape[0] == ref_rois.shape[0]
    dd = np.abs(rois[:, None, 1:] - ref_rois[None, :, 1:]).max(axis=2).min(axis=1)
    assert (dd < 0.05).mean() > 0.95, 'only %.1f%% of device tubes found in the oracle set' % (100 * (dd < 0.05).mean())
    # per-RoI res5 head on the DEVICE tubes (max_box_rois: the oracle's per-RoI res5 on the first rows only)
    nb = rois.shape[0] if max_box_rois is None else min(max_box_rois, rois.shape[0])
    cls_prob, bbox_pred = net.box_head_c4_tube(feat, rois[:nb])
    np.testing.assert_allclose(ws.FetchBlob('cls_prob')[:nb], cls_prob, atol=1e-4)
    got_bp = ws.FetchBlob('bbox_pred')
    assert got_bp.shape == (rois.shape[0], 2 * T * 4) and bbox_pred.shape == (nb, 2 * T * 4)
    np.testing.assert_allclose(got_bp[:nb], bbox_pred, atol=1e-3)
    # 3D keypoint head
    kp_rois = rois[:n_kp].copy()
    ws.FeedBlob('keypoint_rois', kp_rois)
    ws.RunNet(model.keypoint_net.name)
    kps = ws.FetchBlob('kps_score')
    ref = net.kps_head_tube(feat, kp_rois).numpy()
    assert kps.shape == ref.shape == (n_kp, T * 17, 56, 56)
    err = np.abs(kps - ref).max()
    print('tube kps_score max-abs %.3e (ref max %.2f)' % (err, np.abs(ref).max()))
    assert err < 1e-3
    return err
