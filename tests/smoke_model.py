"""smoke(): ONE tiny forward of the detector hot path on cuda:0 (body + FPN3D + RPN/proposals + box head +
keypoint head, fp32 parity mode) checked against the CPU oracle.  Used by __graft_entry__.smoke()."""
import numpy as np
import torch

from tests.model_util import fpn3d_kps_cfg, build_product, synthetic_clip, oracle_opts


def run_smoke(T=2, H=64, W=96):
    from oracle.net3d import Net
    model, ws, weights = build_product(fpn3d_kps_cfg('18', T=T, pre=200, post=50, dtype='fp32'))
    data = synthetic_clip(T, H, W)
    im_info = np.array([[H, W, 1.0]], dtype=np.float32)
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', im_info)
    ws.RunNet(model.net.name)
    rois = ws.FetchBlob('rois')
    assert rois.shape[0] > 0 and rois.shape[1] == 5
    kp_rois = rois[:6].copy()
    ws.FeedBlob('keypoint_rois', kp_rois)
    ws.RunNet(model.keypoint_net.name)
    kps = ws.FetchBlob('kps_score')
    torch.cuda.synchronize()
    # oracle on the same weights / inputs / rois
    net = Net(weights, oracle_opts('18', T, 3, 'slice-center', 200, 50))
    net.body(torch.from_numpy(data))
    p2d = net.time_link(net.fpn())
    from oracle import proposals as op
    _, per_level, restore = op.distribute(kp_rois, 2, 5)
    feat = net.roi_feat_fpn(p2d[1:], per_level, restore, 14, 2)
    ref = net.kps_head_2d(feat).numpy()
    err = float(np.abs(kps - ref).max())
    assert kps.shape == ref.shape == (kp_rois.shape[0], 17, 56, 56)
    assert err < 1e-3, 'kps_score max-abs error %.3e exceeds the 1e-3 fp32 parity bar' % err
    return err
