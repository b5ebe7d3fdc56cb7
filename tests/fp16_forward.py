"""Helper of tests/test_gpu_fp16.py, run as a script in a process started with DAT_H16=fp16 (the 16-bit format is a property of the loaded
library build): the R-18 FPN3D keypoint model in cfg.HIP.DTYPE 'fp16' on one synthetic clip against the ORACLE graph (torch-CPU fp32) --
relative error of body / FPN blobs, proposal agreement, kps_score error -- and the same figures for nothing else; prints one JSON line."""
import json
import sys

import numpy as np
import torch


def main():
    from tests.model_util import fpn3d_kps_cfg, build_product, synthetic_clip, oracle_opts
    from oracle.net3d import Net
    from oracle import proposals as op
    from detectandtrack_amd import libdat
    from detectandtrack_amd.ops import hip_ops as ops
    T, H, W = 4, 96, 128
    model, ws, weights = build_product(fpn3d_kps_cfg('18', T=T, dtype='fp16'))
    data = synthetic_clip(T, H, W)
    im_info = np.array([[H, W, 1.0]], dtype=np.float32)
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', im_info)
    ws.RunNet(model.net.name)
    net = Net(weights, oracle_opts('18', T, 3, 'slice-center', 300, 100))
    net.body(torch.from_numpy(data))
    pyr = net.fpn()
    out = {'h16_format': int(libdat.lib().dat_h16_format()), 'tensor_dtype': str(ws.blobs['res2_1_sum'].t.dtype), 'blobs': {}}
    for n in ('pool1', 'res2_1_sum', 'res3_1_sum', 'res4_1_sum', 'res5_1_sum', 'fpn_res5_1_sum', 'fpn_res2_1_sum'):
        got, ref = ws.FetchBlob(n), net.blobs[n].numpy()
        out['blobs'][n] = float(np.abs(got - ref).max() / np.abs(ref).max())
    p2d = net.time_link(pyr)
    ref_rois, _, _ = net.fpn_rpn(p2d, im_info)
    rois = ws.FetchBlob('rois')
    d = np.abs(rois[:, None, 1:] - ref_rois[None, :, 1:]).max(axis=2).min(axis=1)
    out['rois'] = [int(rois.shape[0]), int(ref_rois.shape[0])]
    out['rois_found_within_1px'] = float((d < 1.0).mean())
    kp_rois = ref_rois[:8].copy()
    ws.FeedBlob('keypoint_rois', kp_rois)
    ws.RunNet(model.keypoint_net.name)
    kps = ws.FetchBlob('kps_score')
    _, per_level, restore = op.distribute(kp_rois, 2, 5)
    ref = net.kps_head_2d(net.roi_feat_fpn(p2d[1:], per_level, restore, 14, 2)).numpy()
    out['kps_max_abs_err'] = float(np.abs(kps - ref).max())
    out['kps_ref_max_abs'] = float(np.abs(ref).max())
    out['kps_argmax_identical'] = float((kps.reshape(8 * 17, -1).argmax(1) == ref.reshape(8 * 17, -1).argmax(1)).mean())
    # the bf16x3 mode lives in the bf16 build only: the fp16 build must refuse it, loudly
    try:
        ops.split_bf16x2(torch.zeros((4, 64), dtype=torch.float32, device='cuda'))
        out['x3_refused'] = False
    except Exception as e:   # noqa: BLE001
        out['x3_refused'] = 'bf16 build' in str(e)
    print(json.dumps(out))


if __name__ == '__main__':
    sys.exit(main())
