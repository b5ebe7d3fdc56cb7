"""The reference's SHIPPED configurations run on the MI355X as shipped: for each of the 12 yaml files of
/root/reference/configs/video -- taken from tests/golden/reference_cfg_files.json, the effective cfg the REAL lib/core/config.py computes
for the file (the GPU box has no /root/reference) -- `model_builder.create` builds the model the file names, one synthetic clip goes through
it at the file's own TEST.SCALES / TEST.MAX_SIZE / VIDEO.NUM_FRAMES and proposal counts (a 1280 x 720 frame through the scale rule of
lib/utils/blob.py:40-90), fp32 parity mode, and every fetched blob and `kps_score` is compared with the oracle graph (< 1e-3).

Three graph families:
  2d_best/01_R101_best_hungarian[-4GPU]   R-101 FPN3D body with T = 1 / kT = 1, 'slice-center', 2-MLP box head, 2D keypoint head -- the
                                          model the reference publishes accuracy for (README.md:157-159): 750 x 1333 padded to 768 x 1344
  3d/01_R-18_*, 02_R-18_*                 2D ResNet-18 C4 body, single-level RPN, per-RoI res5 head, 2D keypoint head: 187 x 333
  3d/03_R-18-3D_*, 04_R-18-3D_*           3D ResNet-18 C4 body (kT = 3, T = 3), tube RPN / RoIAlign, 3D keypoint head: 3 x 187 x 333
"""
import json
import os

import numpy as np
import pytest
import torch

from tests.model_util import synthetic_clip, check_c4_tube_against_oracle

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(REPO, 'tests', 'golden', 'reference_cfg_files.json')) as _f:
    SHIPPED = json.load(_f)


def _build_from_fixture(rel):
    from detectandtrack_amd.core.config import cfg, reset_cfg
    from detectandtrack_amd.modeling import model_builder
    from detectandtrack_amd.utils import net as net_utils
    from detectandtrack_amd import workspace
    reset_cfg()
    for key, v in SHIPPED[rel].items():
        d, parts = cfg, key.split('.')
        for p in parts[:-1]:
            d = d[p]
        d[parts[-1]] = tuple(v) if isinstance(d[parts[-1]], tuple) and isinstance(v, list) else v
    cfg.HIP.DTYPE = 'fp32'
    model = model_builder.create(cfg.MODEL.TYPE, train=False)
    workspace.ResetWorkspace()
    ws = workspace.GlobalWorkspace()
    weights = net_utils.synthetic_params(model, 3)
    for k, v in weights.items():
        ws.set_param(k, v)
    for net in (model.net, model.conv_body_net, model.keypoint_net):
        ws.CreateNet(net)
    return cfg, model, ws, weights


def _geometry(cfg):
    """blob size of a 1280 x 720 frame: scale = min(SCALE / short side, MAX_SIZE / long side) (utils/blob.py:66-90), FPN models padded to
    a multiple of FPN.COARSEST_STRIDE (:40-63)"""
    s = min(float(cfg.TEST.SCALES[0]) / 720., float(cfg.TEST.MAX_SIZE) / 1280.)
    h, w = int(np.rint(720 * s)), int(np.rint(1280 * s))
    if cfg.FPN.FPN_ON:
        st = float(cfg.FPN.COARSEST_STRIDE)
        h, w = int(np.ceil(h / st) * st), int(np.ceil(w / st) * st)
    return h, w, s


def _close(name, got, ref):
    ref = ref.numpy() if isinstance(ref, torch.Tensor) else ref
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err, mx = float(np.abs(got - ref).max()), float(np.abs(ref).max())
    print('%-28s max-abs %.3e (ref max %.2f)' % (name, err, mx))
    assert err < 1e-3 * max(1.0, mx), (name, err, mx)


def _rois_found(rois, ref_rois):
    assert rois.shape == ref_rois.shape, (rois.shape, ref_rois.shape)
    d = np.abs(rois[:, None, 1:] - ref_rois[None, :, 1:]).max(axis=2).min(axis=1)
    assert (d < 0.05).mean() > 0.95, 'only %.1f%% of device rois found in the oracle set' % (100 * (d < 0.05).mean())


def _check_r101_fpn(cfg, model, ws, weights):
    from oracle.net3d import Net, opts_for
    from oracle import proposals as op
    assert cfg.MODEL.CONV_BODY == 'FPN3D.add_fpn_ResNet101_conv5_body' and cfg.VIDEO.NUM_FRAMES == 1 and cfg.VIDEO.BODY_HEAD_LINK == 'slice-center'
    assert cfg.VIDEO.TIME_KERNEL_DIM.BODY == 1
    H, W, s = _geometry(cfg)
    assert (H, W) == (768, 1344)
    data = synthetic_clip(1, H, W)
    im_info = np.array([[H, W, s]], dtype=np.float32)
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', im_info)
    ws.RunNet(model.net.name)
    pre, post = cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    net = Net(weights, opts_for('R101', kt_body=1, body_head_link='slice-center', num_frames_mid=1, pre_nms_topn=pre, post_nms_topn=post))
    net.body(torch.from_numpy(data))
    pyr = net.fpn()
    for n in ('pool1', 'res2_2_sum', 'res3_3_sum', 'res4_22_sum', 'res5_2_sum', 'fpn_res5_2_sum', 'fpn_res4_22_sum', 'fpn_res3_3_sum',
              'fpn_res2_2_sum'):
        _close(n, ws.FetchBlob(n), net.blobs[n])
    p2d = net.time_link(pyr)
    ref_rois, _, _ = net.fpn_rpn(p2d, im_info)
    for lvl in range(2, 7):
        head = ws.FetchBlob('rpn_cls_logits_fpn%d+rpn_bbox_pred_fpn%d' % (lvl, lvl))
        np.testing.assert_allclose(1.0 / (1.0 + np.exp(-head[:, :3])), net.blobs['rpn_cls_probs_fpn%d' % lvl].numpy(), atol=1e-4)
        np.testing.assert_allclose(head[:, 3:15], net.blobs['rpn_bbox_pred_fpn%d' % lvl].numpy(), atol=1e-3)
    rois = ws.FetchBlob('rois')
    assert rois.shape[1] == 5
    _rois_found(rois, ref_rois)
    _, per_level, restore = op.distribute(rois, 2, 5)
    cls_prob, bbox_pred = net.box_head_2mlp(net.roi_feat_fpn(p2d[1:], per_level, restore, 7, 2))
    np.testing.assert_allclose(ws.FetchBlob('cls_prob'), cls_prob, atol=1e-4)
    np.testing.assert_allclose(ws.FetchBlob('bbox_pred'), bbox_pred, atol=1e-3)
    kp_rois = rois[:8].copy()
    ws.FeedBlob('keypoint_rois', kp_rois)
    ws.RunNet(model.keypoint_net.name)
    _, per_level, restore = op.distribute(kp_rois, 2, 5)
    ref = net.kps_head_2d(net.roi_feat_fpn(p2d[1:], per_level, restore, 14, 2)).numpy()
    kps = ws.FetchBlob('kps_score')
    assert kps.shape == ref.shape == (8, 17, 56, 56)
    err = float(np.abs(kps - ref).max())
    print('R-101-FPN kps_score max-abs %.3e (ref max %.2f)' % (err, np.abs(ref).max()))
    assert err < 1e-3


def _check_r18_c4_2d(cfg, model, ws, weights):
    from oracle.net3d import Net, opts_for
    from tests.model_util import oracle_weights_2d
    assert cfg.MODEL.CONV_BODY == 'ResNet.add_ResNet18_conv4_body' and not cfg.MODEL.VIDEO_ON and not cfg.FPN.FPN_ON
    H, W, s = _geometry(cfg)
    assert (H, W) == (187, 333)
    data = synthetic_clip(1, H, W)[:, :, 0]                                  # the 2D blob: 1 x 3 x H x W
    im_info = np.array([[H, W, s]], dtype=np.float32)
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', im_info)
    ws.RunNet(model.net.name)
    pre, post = cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N
    net = Net(oracle_weights_2d(weights), opts_for('R18', block_counts=(2, 2, 2), kt_body=1, body_head_link='', num_frames_mid=1,
                                                   pre_nms_topn=pre, post_nms_topn=post, rpn_sizes=tuple(cfg.RPN.SIZES),
                                                   rpn_c4_aspect_ratios=tuple(cfg.RPN.ASPECT_RATIOS)))
    feat = net.body(torch.from_numpy(data[:, :, None]))[:, :, 0]
    for n in ('pool1', 'res2_1_sum', 'res3_1_sum', 'res4_1_sum'):
        _close(n, ws.FetchBlob(n), net.blobs[n][:, :, 0])
    ref_rois, _ = net.rpn_c4_2d(feat, im_info)[:2]
    A = 12
    head = ws.FetchBlob('rpn_cls_logits+rpn_bbox_pred')                       # (1, 5A, h, w)
    assert head.shape[1] == 5 * A
    np.testing.assert_allclose(1.0 / (1.0 + np.exp(-head[:, :A])), net.blobs['rpn_cls_probs'].numpy(), atol=1e-4)
    np.testing.assert_allclose(head[:, A:], net.blobs['rpn_bbox_pred'].numpy(), atol=1e-3)
    rois = ws.FetchBlob('rois')
    assert rois.shape[1] == 5
    _rois_found(rois, ref_rois)
    nb = min(200, rois.shape[0])                                              # (the oracle's per-RoI res5 on the first rows)
    cls_prob, bbox_pred = net.box_head_c4_2d(feat, rois[:nb])
    np.testing.assert_allclose(ws.FetchBlob('cls_prob')[:nb], cls_prob, atol=1e-4)
    np.testing.assert_allclose(ws.FetchBlob('bbox_pred')[:nb], bbox_pred, atol=1e-3)
    kp_rois = rois[:8].copy()
    ws.FeedBlob('keypoint_rois', kp_rois)
    ws.RunNet(model.keypoint_net.name)
    ref = net.kps_head_c4_2d(feat, kp_rois).numpy()
    kps = ws.FetchBlob('kps_score')
    assert kps.shape == ref.shape == (8, 17, 56, 56)
    err = float(np.abs(kps - ref).max())
    print('R-18 C4 2D kps_score max-abs %.3e (ref max %.2f)' % (err, np.abs(ref).max()))
    assert err < 1e-3


def _check_r18_c4_tube(cfg, model, ws, weights):
    assert cfg.MODEL.CONV_BODY == 'ResNet3D.add_ResNet18_conv4_body' and cfg.MODEL.VIDEO_ON and cfg.VIDEO.NUM_FRAMES == 3
    assert cfg.KRCNN.NO_3D_DECONV_TIME_TO_CH and cfg.VIDEO.BODY_HEAD_LINK == ''
    H, W, s = _geometry(cfg)
    assert (H, W) == (187, 333)
    check_c4_tube_against_oracle(model, ws, weights, 3, H, W, cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N, im_scale=s, n_kp=8,
                                 max_box_rois=200)


@pytest.mark.parametrize('rel', sorted(SHIPPED))
def test_shipped_config_forward_matches_the_oracle(rel):
    from detectandtrack_amd.core.config import reset_cfg
    try:
        cfg, model, ws, weights = _build_from_fixture(rel)
        if 'R101' in rel:
            _check_r101_fpn(cfg, model, ws, weights)
        elif '-3D_' in rel:
            _check_r18_c4_tube(cfg, model, ws, weights)
        else:
            _check_r18_c4_2d(cfg, model, ws, weights)
    finally:
        reset_cfg()
        from detectandtrack_amd import workspace
        workspace.ResetWorkspace()
