"""What the bf16 performance mode costs in accuracy, measured against the fp32 parity mode of the SAME model and weights
(bench.py prints this next to `dtype`; tests/test_gpu_parity_full.py gates it).

The north-star bar (heatmaps within 1e-3 max-abs) is stated for fp32 and is checked against the oracle in the parity tests;
this module answers the other question: how far the benched bf16 numbers are from that fp32 path on the benched clip --
`kps_score` max-abs error, identical arg-max cells, decoded keypoints within 1 px, and agreement of the proposal sets.
Both sides are product code (two workspaces, one per arithmetic mode); no oracle is involved.
"""
import numpy as np
import torch

from detectandtrack_amd.core.config import cfg
from detectandtrack_amd.ops import hip_ops as ops
from detectandtrack_amd import workspace as wsmod


def second_workspace(ws, model, dtype):
    """A workspace with the same parameters and nets as `ws` running in `dtype` ('fp32' | 'bf16')."""
    w = wsmod.Workspace(ws.device.index, dtype=dtype)
    for k, v in ws.params.items():
        w.set_param(k, v)
    for net in (model.net, model.conv_body_net, model.keypoint_net):
        if net is not None:
            w.CreateNet(net)
    return w


def detect(model, ws, data, im_info):
    """model.net on one resident clip: rois, class probabilities, box deltas (host arrays)."""
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', im_info)
    ws.RunNet(model.net.name)
    return ws.FetchBlob('rois'), ws.FetchBlob('cls_prob'), ws.FetchBlob('bbox_pred')


def keypoints(model, ws, kp_rois, im_scale):
    """keypoint_net on `kp_rois` (R x (4T+1), network coordinates): kps_score (device fp32 tensor) and the decoded rows
    R x 4 x 17T (x, y in image coordinates, logit, prob)."""
    ws.FeedBlob('keypoint_rois', np.ascontiguousarray(kp_rois, dtype=np.float32))
    ws.RunNet(model.keypoint_net.name)
    heat = ws.blobs['kps_score'].t
    T = (kp_rois.shape[1] - 1) // 4
    boxes = torch.from_numpy(np.ascontiguousarray(kp_rois[:, 1:] / im_scale, dtype=np.float32)).to(heat.device)
    xy = ops.heatmaps_to_keypoints(heat.contiguous(), boxes, T, cfg.KRCNN.NUM_KEYPOINTS, cfg.KRCNN.INFERENCE_MIN_SIZE)
    return heat, xy.cpu().numpy()


def set_agreement(a, b, tol):
    """Fraction of the rows of `a` that have a row of `b` within `tol` (max-abs over the coordinates)."""
    if a.shape[0] == 0 or b.shape[0] == 0:
        return 0.0
    ta, tb = torch.from_numpy(np.ascontiguousarray(a)), torch.from_numpy(np.ascontiguousarray(b))
    hit = 0
    for i in range(0, ta.shape[0], 256):
        d = (ta[i:i + 256, None, :] - tb[None, :, :]).abs().amax(dim=2).amin(dim=1)
        hit += int((d < tol).sum())
    return hit / float(ta.shape[0])


def best_iou(a, b):
    """For every box of `a` (n x 4) the best IoU (+1 pixel convention of lib/utils/boxes.py) with a box of `b`."""
    if a.shape[0] == 0 or b.shape[0] == 0:
        return np.zeros((a.shape[0],), np.float32)
    ta, tb = torch.from_numpy(np.ascontiguousarray(a[:, :4])).double(), torch.from_numpy(np.ascontiguousarray(b[:, :4])).double()
    area_b = (tb[:, 2] - tb[:, 0] + 1) * (tb[:, 3] - tb[:, 1] + 1)
    out = []
    for i in range(0, ta.shape[0], 256):
        x = ta[i:i + 256]
        area_a = (x[:, 2] - x[:, 0] + 1) * (x[:, 3] - x[:, 1] + 1)
        w = (torch.minimum(x[:, None, 2], tb[None, :, 2]) - torch.maximum(x[:, None, 0], tb[None, :, 0]) + 1).clamp(min=0)
        h = (torch.minimum(x[:, None, 3], tb[None, :, 3]) - torch.maximum(x[:, None, 1], tb[None, :, 1]) + 1).clamp(min=0)
        inter = w * h
        out.append((inter / (area_a[:, None] + area_b[None, :] - inter)).amax(dim=1))
    return torch.cat(out).numpy()


def blob_error_trail(net, ws_a, ws_b, abs_bar=1e-2):
    """Where the arithmetic of workspace `ws_a` (bf16) leaves that of `ws_b` (fp32): for every feature-map / row blob both hold
    after running `net`, in the order the net produces them, the max-abs difference, the reference magnitude and the mean-abs
    difference -- and the first blob whose max-abs difference exceeds `abs_bar`.  Everything is computed on the device."""
    trail, first = [], None
    seen = set()
    for op in net.ops:
        for name in op.outputs:
            a, b = ws_a.blobs.get(name), ws_b.blobs.get(name)
            if name in seen or a is None or b is None or a.kind not in ('fmap', 'rows') or a.kind != b.kind:
                continue
            seen.add(name)
            if tuple(a.t.shape) != tuple(b.t.shape):
                continue
            C = a.C
            ta, tb = a.t[..., :C].float(), b.t[..., :C].float()
            d = (ta - tb).abs()
            rec = {'blob': name, 'max_abs_err': float(d.max()), 'mean_abs_err': float(d.mean()), 'ref_max_abs': float(tb.abs().max()),
                   'ref_mean_abs': float(tb.abs().mean())}
            del d, ta, tb
            trail.append(rec)
            if first is None and rec['max_abs_err'] > abs_bar:
                first = rec
    return trail, first


def bf16_vs_fp32(model, ws_bf16, data, im_info, n_kp=100, ws_fp32=None, trail=False):
    """Run the clip through the bf16 workspace and an fp32 twin; both keypoint nets get the SAME rois (the fp32 path's
    best-scoring boxes).  Returns a flat dict of error figures."""
    ws32 = ws_fp32 or second_workspace(ws_bf16, model, 'fp32')
    r32, p32, _ = detect(model, ws32, data, im_info)
    r16, p16, _ = detect(model, ws_bf16, data, im_info)
    out = {
        'rois_fp32': int(r32.shape[0]), 'rois_bf16': int(r16.shape[0]),
    }
    if trail:
        # which blob first leaves 1e-2: the bf16 operand rounding of the INPUT (|x| <= 152 -> half an ulp = 0.5) is
        # already > 1e-2 absolute at conv1, so the trail is reported with the magnitudes next to it: what matters is err / |ref|
        tr, first = blob_error_trail(model.net, ws_bf16, ws32)
        out['first_blob_over_1e-2'] = {k: (v if isinstance(v, str) else round(v, 5)) for k, v in first.items()} if first else None
        rel = [(r['blob'], r['max_abs_err'] / max(r['ref_max_abs'], 1e-30)) for r in tr]
        out['blob_rel_err_trail'] = {n: round(v, 5) for n, v in rel if n.endswith('_sum') or n in ('pool1', 'conv1', 'fc6', 'fc7')}
        worst = max(rel, key=lambda kv: kv[1]) if rel else None
        out['worst_blob_rel_err'] = {'blob': worst[0], 'max_abs_err_over_ref_max': round(worst[1], 5)} if worst else None
    iou = best_iou(r16[:, 1:5], r32[:, 1:5])
    # proposals: the share of bf16 rois that have an fp32 roi of IoU >= 0.9 / 0.7 (bf16 noise in the box deltas moves a
    # 256-px anchor by a few px, and near-tied objectness scores swap places at the top-N cut: a set comparison, not a row one)
    out.update({'rois_matched_iou_0.9': round(float((iou >= 0.9).mean()), 4), 'rois_matched_iou_0.7': round(float((iou >= 0.7).mean()), 4),
                'rois_mean_best_iou': round(float(iou.mean()), 4)})
    order = np.argsort(-p32[:, 1], kind='stable')[:n_kp]
    kp_rois = r32[order]
    scale = float(np.asarray(im_info).reshape(-1)[2])
    h32, xy32 = keypoints(model, ws32, kp_rois, scale)
    h16, xy16 = keypoints(model, ws_bf16, kp_rois, scale)
    R, C, M, _ = h32.shape
    err = (h16 - h32).abs()
    a32, a16 = h32.view(R, C, -1).argmax(dim=2), h16.view(R, C, -1).argmax(dim=2)
    dist = np.sqrt((xy16[:, 0] - xy32[:, 0]) ** 2 + (xy16[:, 1] - xy32[:, 1]) ** 2)
    out.update({
        'kps_rois': int(R),
        'kps_score_max_abs_err': round(float(err.max()), 5),
        'kps_score_mean_abs_err': round(float(err.mean()), 6),
        'kps_score_ref_max_abs': round(float(h32.abs().max()), 4),
        'kps_argmax_cell_identical': round(float((a32 == a16).float().mean()), 4),
        'keypoints_within_1px': round(float((dist <= 1.0).mean()), 4),
        'keypoint_mean_px_err': round(float(dist.mean()), 4),
    })
    return out
