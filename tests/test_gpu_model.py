"""End-to-end GPU parity: the recorded graph executed through the C ABI vs the oracle graph (fp32 parity mode,
north-star bar: kps_score within 1e-3 max-abs) plus the bf16 performance mode with its own (looser) bound."""
import os

import numpy as np
import pytest
import torch

from tests.model_util import fpn3d_kps_cfg, build_product, synthetic_clip, oracle_opts

pytestmark = pytest.mark.gpu


def _oracle_pyramid(weights, arch, data, T, kt=3, link='slice-center', pre=300, post=100):
    from oracle.net3d import Net
    net = Net(weights, oracle_opts(arch, T, kt, link, pre, post))
    net.body(torch.from_numpy(data))
    pyr = net.fpn()
    return net, pyr


@pytest.mark.parametrize('arch,T,H,W', [('18', 4, 96, 128), ('50', 2, 64, 96)])
def test_fp32_forward_matches_oracle(arch, T, H, W):
    from oracle import proposals as op
    model, ws, weights = build_product(fpn3d_kps_cfg(arch, T=T, dtype='fp32'))
    data = synthetic_clip(T, H, W)
    im_info = np.array([[H, W, 1.0]], dtype=np.float32)
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', im_info)
    ws.RunNet(model.net.name)
    net, pyr = _oracle_pyramid(weights, arch, data, T)
    last = {'18': 'res%d_1_sum', '50': None}
    # body + FPN blobs (NC(T)HW fp32 at the boundary)
    names = ['pool1'] + [b for b in ws.Blobs() if b.endswith('_sum') and b.startswith('res')] + \
            [b for b in ws.Blobs() if b.startswith('fpn_res') and b.endswith('_sum')]
    for n in names:
        got = ws.FetchBlob(n)
        ref = net.blobs[n].numpy()
        assert got.shape == ref.shape, (n, got.shape, ref.shape)
        err = np.abs(got - ref).max()
        print('%-24s max-abs %.3e (ref max %.2f)' % (n, err, np.abs(ref).max()))
        assert err < 1e-3 * max(1.0, np.abs(ref).max()), n
    # RPN head outputs, per level
    p2d = net.time_link(pyr)
    ref_rois, _, _ = net.fpn_rpn(p2d, im_info)
    for lvl in range(2, 7):
        head = ws.FetchBlob('rpn_cls_logits_fpn%d+rpn_bbox_pred_fpn%d' % (lvl, lvl))
        probs = 1.0 / (1.0 + np.exp(-head[:, :3]))
        np.testing.assert_allclose(probs, net.blobs['rpn_cls_probs_fpn%d' % lvl].numpy(), atol=1e-4)
        np.testing.assert_allclose(head[:, 3:15], net.blobs['rpn_bbox_pred_fpn%d' % lvl].numpy(), atol=1e-3)
    # proposals: same set up to fp32 re-ordering of near-equal scores
    rois = ws.FetchBlob('rois')
    assert rois.shape[1] == 5 and rois.shape[0] == ref_rois.shape[0]
    d = np.abs(rois[:, None, 1:] - ref_rois[None, :, 1:]).max(axis=2).min(axis=1)
    assert (d < 0.05).mean() > 0.95, 'only %.1f%% of device rois found in the oracle set' % (100 * (d < 0.05).mean())
    # box head on the DEVICE rois (oracle features, oracle head)
    _, per_level, restore = op.distribute(rois, 2, 5)
    feat = net.roi_feat_fpn(p2d[1:], per_level, restore, 7, 2)
    cls_prob, bbox_pred = net.box_head_2mlp(feat)
    np.testing.assert_allclose(ws.FetchBlob('cls_prob'), cls_prob, atol=1e-4)
    np.testing.assert_allclose(ws.FetchBlob('bbox_pred'), bbox_pred, atol=1e-3)
    # keypoint net on a few boxes
    kp_rois = rois[:7].copy()
    ws.FeedBlob('keypoint_rois', kp_rois)
    ws.RunNet(model.keypoint_net.name)
    kps = ws.FetchBlob('kps_score')
    _, per_level, restore = op.distribute(kp_rois, 2, 5)
    ref = net.kps_head_2d(net.roi_feat_fpn(p2d[1:], per_level, restore, 14, 2)).numpy()
    err = np.abs(kps - ref).max()
    print('kps_score max-abs %.3e (ref max %.2f)' % (err, np.abs(ref).max()))
    assert kps.shape == ref.shape
    assert err < 1e-3


def test_bf16_forward_close_to_oracle():
    """Performance mode: bf16 activations/weights with fp32 accumulation; reported, looser bound."""
    T, H, W = 4, 96, 128
    model, ws, weights = build_product(fpn3d_kps_cfg('18', T=T, dtype='bf16'))
    data = synthetic_clip(T, H, W)
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', np.array([[H, W, 1.0]], dtype=np.float32))
    ws.RunNet(model.net.name)
    net, pyr = _oracle_pyramid(weights, '18', data, T)
    for n in ('res2_1_sum', 'res5_1_sum', 'fpn_res2_1_sum'):
        got, ref = ws.FetchBlob(n), net.blobs[n].numpy()
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        print('bf16 %-16s max-abs/max %.3e' % (n, rel))
        assert rel < 0.06, n
    rois = ws.FetchBlob('rois')
    assert rois.shape[0] > 0
    ws.FeedBlob('keypoint_rois', rois[:5].copy())
    ws.RunNet(model.keypoint_net.name)
    assert np.isfinite(ws.FetchBlob('kps_score')).all()


def test_im_detect_all_surface():
    """The reference engine surface (core/test.py:897) runs end to end on synthetic frames."""
    from detectandtrack_amd.core import test as test_engine
    from detectandtrack_amd.core.config import cfg
    T = 2
    model, ws, _ = build_product(fpn3d_kps_cfg('18', T=T, dtype='fp32'))
    cfg.TEST.SCALES = (64,)
    cfg.TEST.MAX_SIZE = 128
    cfg.TEST.SCORE_THRESH = 0.0
    rs = np.random.RandomState(0)
    frames = [rs.randint(0, 255, (60, 90, 3)).astype(np.uint8) for _ in range(T)]
    cls_boxes, cls_segms, cls_keyps = test_engine.im_detect_all(model, frames, None)
    assert cls_segms is None
    assert cls_boxes[1].shape[1] == 5 and cls_boxes[1].shape[0] <= cfg.TEST.DETECTIONS_PER_IM
    assert cls_boxes[1].shape[0] > 0
    assert len(cls_keyps[1]) == cls_boxes[1].shape[0] and cls_keyps[1][0].shape == (4, 17)
    # the default runs the glue between the nets (box decode, threshold, NMS, top-100) AND the heatmap decode on the device, with one
    # read-back per clip; the reference's host glue with the device heatmap decode, and the all-host path, must give the same rows
    # (boxes to the 1-ulp difference between NumPy's float32 exp and the correctly rounded one; everything else follows the boxes)
    def same(a_boxes, a_keyps, b_boxes, b_keyps):
        assert a_boxes[1].shape == b_boxes[1].shape
        np.testing.assert_array_equal(a_boxes[1][:, 4], b_boxes[1][:, 4])
        np.testing.assert_allclose(a_boxes[1][:, :4], b_boxes[1][:, :4], rtol=0, atol=2e-3)
        moved = total = 0
        for a, b in zip(a_keyps[1], b_keyps[1]):
            np.testing.assert_allclose(a[2], b[2], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(a[3], b[3], rtol=1e-4)
            moved += int((np.abs(a[:2] - b[:2]).max(axis=0) > 5e-3).sum())
            total += a.shape[1]
        # (the two glue paths hand the keypoint net rois that differ by an ulp -- see above; a heatmap whose two best bins tie to 1e-6
        #  may then resolve to the other bin: the logits above agree, the position moves.  tools/probes/tie_probe.py: each path alone
        #  is bit-reproducible run to run.)
        assert moved <= max(1, total // 200), (moved, total)
    cfg.HIP.DEVICE_BOX_RESULTS = False
    cls_boxes_g, _, cls_keyps_g = test_engine.im_detect_all(model, frames, None)
    same(cls_boxes, cls_keyps, cls_boxes_g, cls_keyps_g)
    cfg.HIP.DEVICE_KPS_DECODE = False
    cls_boxes_h, _, cls_keyps_h = test_engine.im_detect_all(model, frames, None)
    np.testing.assert_array_equal(cls_boxes_g[1], cls_boxes_h[1])
    for a, b in zip(cls_keyps_g[1], cls_keyps_h[1]):
        np.testing.assert_array_equal(a[:3], b[:3])
        np.testing.assert_allclose(a[3], b[3], rtol=2e-5)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_clip_graph_replay_equals_eager(dtype):
    """core/clip_graph.ClipGraph: a clip captured as one hipGraph (model.net + device post-processing + keypoint net + decode) and
    replayed on new clips gives exactly the eager results -- same kernels, same arguments."""
    from detectandtrack_amd.core import test as engine
    from detectandtrack_amd.core.clip_graph import ClipGraph
    from detectandtrack_amd.core.config import cfg
    T, H, W = 4, 96, 128
    model, ws, _ = build_product(fpn3d_kps_cfg('18', T=T, dtype=dtype))
    cfg.TEST.SCORE_THRESH = 0.0
    im_info = np.array([[H, W, 1.0]], dtype=np.float32)
    clips = [torch.from_numpy(synthetic_clip(T, H, W, seed=s)).cuda() for s in (3, 4, 5)]

    def eager(data):
        ws.FeedBlob('data', data)
        ws.FeedBlob('im_info', im_info)
        ws.RunNet(model.net.name)
        return engine.read_results_from_device(*engine.enqueue_results_on_device(model, (H, W, 3), 1.0))
    ref = [eager(c) for c in clips]
    g = ClipGraph(model, ws, clips[0], im_info, (H, W, 3), stream=torch.cuda.Stream())
    for c, (rb, rk) in zip(clips, ref):
        g.launch(c)
        boxes, keyps = g.results()
        np.testing.assert_array_equal(boxes[1], rb[1])
        assert len(keyps[1]) == len(rk[1]) > 0
        for a, b in zip(keyps[1], rk[1]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_frame_trunk_cache_gives_identical_sliding_window_results(dtype):
    """cfg.HIP.FRAME_TRUNK_CACHE: conv1 / pool1 / res2 have no temporal extent, so a sliding window (one clip per key frame,
    stride 1, border frames replicated, reference utils/video.py:149-201) re-uses their per-frame output.  Detections and
    keypoints of every clip must be IDENTICAL to the plain path, while only the new frame of each clip runs the trunk."""
    from detectandtrack_amd.core import test as test_engine
    from detectandtrack_amd.core.config import cfg
    from detectandtrack_amd.workspace import Executor
    T, n_frames = 4, 7
    model, ws, _ = build_product(fpn3d_kps_cfg('18', T=T, dtype=dtype))
    cfg.TEST.SCALES = (64,)
    cfg.TEST.MAX_SIZE = 128
    cfg.TEST.SCORE_THRESH = 0.0
    assert Executor.trunk_split(model.net) == (7, 'res2_1_sum')
    rs = np.random.RandomState(0)
    video = [rs.randint(0, 255, (60, 90, 3)).astype(np.uint8) for _ in range(n_frames)]
    clips = []
    for key in range(n_frames):          # clip around every key frame, border frames replicated
        ids = [min(max(key - T // 2 + j, 0), n_frames - 1) for j in range(T)]
        clips.append(ids)
    plain = [test_engine.im_detect_all(model, [video[i] for i in ids], None) for ids in clips]
    cfg.HIP.FRAME_TRUNK_CACHE = 6
    stem_frames = []
    orig = Executor._stem

    def counting_stem(self, i, op, xin):
        stem_frames.append(int(xin.t.shape[2]))
        return orig(self, i, op, xin)
    Executor._stem = counting_stem
    try:
        cached = [test_engine.im_detect_all(model, [video[i] for i in ids], None, frame_ids=[('vid0', i) for i in ids]) for ids in clips]
    finally:
        Executor._stem = orig
        cfg.HIP.FRAME_TRUNK_CACHE = 0
    assert sum(stem_frames) == n_frames, stem_frames          # every video frame went through the trunk exactly once
    assert len(ws.trunk_cache) <= 6
    for (b0, _, k0), (b1, _, k1) in zip(plain, cached):
        np.testing.assert_array_equal(b0[1], b1[1])
        assert len(k0[1]) == len(k1[1])
        for a, b in zip(k0[1], k1[1]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('deconv,T', [('time_to_batch', 3), ('grouped', 3), ('group_ignored', 3), ('grouped', 2)])
def test_c4_tube_forward_matches_oracle(deconv, T):
    """The shipped 3D configuration (configs/video/3d/04_R-18-3D_*.yaml): 3D C4 body -> tube RPN (logits averaged
    over T, per-frame deltas) -> tube RoIAlign -> per-RoI res5 -> T-averaged class scores / per-frame box deltas,
    and the 3D keypoint head with per-frame deconvs (heatmap channels t*K + k) -- with the shipped configs' time -> batch deconv
    (shared weights) and with the reference DEFAULT, KRCNN.NO_3D_DECONV_TIME_TO_CH False (model_builder.py:765-767, :848-868: time ->
    channels, ConvTranspose group = T on T*C -> T*K channels, bilinear deconv over T*K maps), in both readings of `group`."""
    from tests.model_util import c4_tube_kps_cfg, check_c4_tube_against_oracle
    H, W = 96, 128
    pre, post = 300, 60
    model, ws, weights = build_product(c4_tube_kps_cfg(T=T, pre=pre, post=post, deconv=deconv))
    Ck = 512
    assert weights['kps_score_lowres_w'].shape == {'time_to_batch': (Ck, 17, 4, 4), 'grouped': (T * Ck, 17, 4, 4),
                                                   'group_ignored': (T * Ck, T * 17, 4, 4)}[deconv]
    assert weights['kps_score_lowres_b'].shape == ((17,) if deconv == 'time_to_batch' else (T * 17,))
    check_c4_tube_against_oracle(model, ws, weights, T, H, W, pre, post, kps_time_to_ch=deconv != 'time_to_batch')


def test_keyframe_dce_gives_identical_head_inputs():
    """cfg.HIP.KEYFRAME_DCE (opt-in): computing only the centre frame of the FPN outputs that 'slice-center' keeps
    must not change anything the heads read."""
    T, H, W = 4, 96, 128
    outs = []
    for dce in (False, True):
        c = fpn3d_kps_cfg('18', T=T, dtype='fp32')
        c['HIP']['KEYFRAME_DCE'] = dce
        model, ws, _ = build_product(c)
        ws.FeedBlob('data', synthetic_clip(T, H, W))
        ws.FeedBlob('im_info', np.array([[H, W, 1.0]], dtype=np.float32))
        ws.RunNet(model.net.name)
        rois = ws.FetchBlob('rois')
        ws.FeedBlob('keypoint_rois', rois[:6].copy())
        ws.RunNet(model.keypoint_net.name)
        sliced = sorted(b for b in ws.Blobs() if b.endswith('_slicekey'))
        assert len(sliced) == 5
        outs.append((rois, ws.FetchBlob('cls_prob'), ws.FetchBlob('kps_score'), [ws.FetchBlob(b) for b in sliced],
                     ws.FetchBlob('fpn_res2_1_sum').shape))
    a, b = outs
    assert len(a[4]) == 5 and a[4][2] == T      # faithful mode keeps all T frames of P2
    assert len(b[4]) == 4                        # DCE mode: only the centre frame exists
    for x, y in zip(a[3], b[3]):
        np.testing.assert_allclose(x, y, atol=1e-5)
    np.testing.assert_allclose(a[0], b[0], atol=1e-3)
    np.testing.assert_allclose(a[1], b[1], atol=1e-5)
    np.testing.assert_allclose(a[2], b[2], atol=1e-5)


def test_fpn3d_tube_heads_match_oracle():
    """Declared extension: tube RPN on every FPN3D level (the design of the reference's dead
    FPN3D.py:232-330: time -> channels, 2D 1x1 heads over C*T inputs), tube rois through the 2-MLP box head
    (fc6 over T*C*49) and the 3D keypoint head."""
    from oracle.net3d import Net
    from tests.model_util import fpn3d_tube_kps_cfg
    T, H, W = 2, 96, 128
    pre, post = 200, 50
    model, ws, weights = build_product(fpn3d_tube_kps_cfg(T=T, pre=pre, post=post))
    data = synthetic_clip(T, H, W)
    im_info = np.array([[H, W, 1.0]], dtype=np.float32)
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', im_info)
    ws.RunNet(model.net.name)
    net = Net(weights, oracle_opts('18', T, 3, '', pre, post))
    net.body(torch.from_numpy(data))
    pyr = net.fpn()                                   # [P6, P5, P4, P3, P2], 5-D
    ref_rois = net.fpn_rpn_tube(pyr, im_info)
    A = 3
    for lvl in range(2, 7):
        head = ws.FetchBlob('rpn_cls_logits_fpn%d+rpn_bbox_pred_fpn%d' % (lvl, lvl))     # (1, A + 4*T*A, h, w)
        assert head.shape[1] == A + 4 * T * A
        probs = 1.0 / (1.0 + np.exp(-head[:, :A]))
        np.testing.assert_allclose(probs, net.blobs['rpn_cls_probs_fpn%d' % lvl].numpy(), atol=1e-4)
        np.testing.assert_allclose(head[:, A:], net.blobs['rpn_bbox_pred_fpn%d' % lvl].numpy(), atol=1e-3)
    moved = ws.FetchBlob('conv_rpn_timepooled_fpn3')  # time moved into channels: (1, T*C, h, w)
    assert moved.shape[1] == T * 256
    rois = ws.FetchBlob('rois')
    assert rois.shape[1] == 4 * T + 1 and rois.shape[0] == ref_rois.shape[0]
    dd = np.abs(rois[:, None, 1:] - ref_rois[None, :, 1:]).max(axis=2).min(axis=1)
    assert (dd < 0.05).mean() > 0.95, 'only %.1f%% of device tubes found in the oracle set' % (100 * (dd < 0.05).mean())
    # box head on the DEVICE tubes
    feat = net.roi_feat_fpn_tube(pyr[1:], rois, 7, 2)
    cls_prob, bbox_pred = net.box_head_2mlp_tube(feat)
    np.testing.assert_allclose(ws.FetchBlob('cls_prob'), cls_prob, atol=1e-4)
    got_bp = ws.FetchBlob('bbox_pred')
    assert got_bp.shape == bbox_pred.shape == (rois.shape[0], 2 * T * 4)
    np.testing.assert_allclose(got_bp, bbox_pred, atol=1e-3)
    # 3D keypoint head on multi-level tube features
    kp_rois = rois[:5].copy()
    ws.FeedBlob('keypoint_rois', kp_rois)
    ws.RunNet(model.keypoint_net.name)
    kps = ws.FetchBlob('kps_score')
    ref = net.kps_head_tube_feat(net.roi_feat_fpn_tube(pyr[1:], kp_rois, 14, 2)).numpy()
    assert kps.shape == ref.shape == (5, T * 17, 56, 56)
    err = np.abs(kps - ref).max()
    print('FPN tube kps_score max-abs %.3e (ref max %.2f)' % (err, np.abs(ref).max()))
    assert err < 1e-3


def test_bf16_forward_at_bench_size_close_to_oracle():
    """The BENCH configuration end to end (R-18 FPN3D, 8 x 768 x 1344 clip, bf16): body + FPN blobs against the oracle graph run on
    the host at full size (a few seconds), plus a decoded-keypoint sanity pass through the engine path."""
    T, H, W = 8, 768, 1344
    c = fpn3d_kps_cfg('18', T=T, dtype='bf16', pre=1000, post=1000)
    model, ws, weights = build_product(c)
    data = synthetic_clip(T, H, W)
    ws.FeedBlob('data', data)
    ws.FeedBlob('im_info', np.array([[H, W, 1.0]], dtype=np.float32))
    ws.RunNet(model.net.name)
    net, pyr = _oracle_pyramid(weights, '18', data, T, pre=1000, post=1000)
    for n in ('pool1', 'res2_1_sum', 'res3_1_sum', 'res5_1_sum', 'fpn_res5_1_sum', 'fpn_res3_1_sum', 'fpn_res2_1_sum'):
        got, ref = ws.FetchBlob(n), net.blobs[n].numpy()
        assert got.shape == ref.shape, (n, got.shape, ref.shape)
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        print('bench-size bf16 %-16s max-abs/max %.3e' % (n, rel))
        assert rel < 0.06, (n, rel)
    rois = ws.FetchBlob('rois')
    assert rois.shape == (1000, 5) and np.isfinite(rois).all()
    assert (rois[:, 3] >= rois[:, 1]).all() and rois[:, 1:].min() >= 0 and rois[:, 3].max() <= W - 1 and rois[:, 4].max() <= H - 1


def test_fused_stem_pool_is_taken_through_runnet_and_is_bit_identical():
    """with the nets registered the way core/test_engine.initialize_model_from_cfg registers them (net, conv_body_net,
    keypoint_net) the fused conv1+pool1 kernel must be the one that runs (conv_body_net is a clone that produces its own conv1),
    and `pool1` must equal the two-kernel path bit for bit."""
    from detectandtrack_amd.core.config import cfg
    from detectandtrack_amd.ops import hip_ops
    T, H, W = 2, 96, 160
    data = synthetic_clip(T, H, W)
    im_info = np.array([[H, W, 1.0]], dtype=np.float32)
    pools = {}
    for dtype in ('bf16', 'fp32'):
        for fuse in (True, False):
            c = fpn3d_kps_cfg('18', T=T, dtype=dtype)
            c['HIP']['FUSE_STEM_POOL'] = fuse
            model, ws, _ = build_product(c)
            assert model.conv_body_net.name in ws.nets and cfg.HIP.FUSE_STEM_POOL == fuse
            calls = []
            orig = hip_ops.StemConv.pooled

            def counting(self, x, _orig=orig, _calls=calls):
                _calls.append(1)
                return _orig(self, x)
            hip_ops.StemConv.pooled = counting
            try:
                ws.FeedBlob('data', data)
                ws.FeedBlob('im_info', im_info)
                ws.RunNet(model.net.name)
            finally:
                hip_ops.StemConv.pooled = orig
            assert len(calls) == (1 if fuse else 0), (dtype, fuse, calls)
            assert ('conv1' in ws.blobs) == (not fuse)
            pools[(dtype, fuse)] = ws.FetchBlob('pool1')
        np.testing.assert_array_equal(pools[(dtype, True)], pools[(dtype, False)])


def _boxes_agree(a, b, tol):
    """fraction of rows of `a` that have a row of `b` within `tol` (max-abs over the columns)"""
    if len(a) == 0 or len(b) == 0:
        return 1.0 if len(a) == len(b) else 0.0
    d = np.abs(a[:, None, :] - b[None, :, :]).max(axis=2).min(axis=1)
    return float((d < tol).mean())


@pytest.mark.parametrize('kind,B,T,H,W', [('2d', 4, 1, 128, 192), ('3d', 2, 4, 96, 128)])
def test_several_images_per_forward_give_each_image_its_own_results(kind, B, T, H, W):
    """Round 3 (SURVEY §8d config 2 'new build may batch N=8'): B frames (2D R-50-FPN) / B clips (3D R-18 FPN3D)
    through ONE forward -- the N axis of the blobs, the image as a grid dimension of the proposal / detection kernels -- must give
    every image what it gets alone through im_detect_all (reference order of results, lib/core/test.py:897-957): the same
    proposals (`rois` rows of image i, col 0 = i), detections and keypoints.  fp32 parity mode.  ADVICE r3: the comparison is TIGHT --
    with split-K forced off (`dat_conv3d_tune_plan(0, 1)`) every output position sums its K axis in the same order whatever the
    batch size and tile shape, so an image's proposals, detections and keypoints in the batch must EQUAL the ones it gets alone
    (a wrong im_info row or batch index on a minority of rois cannot hide in a tolerance); the planner's own split-K choice is
    then checked at the looser, summation-order tolerance."""
    from detectandtrack_amd.core import test as engine
    from detectandtrack_amd.core.config import cfg
    from detectandtrack_amd.ops import hip_ops
    from tests.model_util import fpn2d_kps_cfg
    c = fpn2d_kps_cfg('50', dtype='fp32', pre=400, post=150) if kind == '2d' else fpn3d_kps_cfg('18', T=T, dtype='fp32', pre=400, post=150)
    c['TEST'].update(SCALES=(H,), MAX_SIZE=max(H, W), SCORE_THRESH=0.0, DETECTIONS_PER_IM=20)
    model, ws, _ = build_product(c)
    rs = np.random.RandomState(11)
    ims = [[rs.randint(0, 255, (H, W, 3)).astype(np.uint8) for _ in range(T)] for _ in range(B)]
    # ---- pass 1: no split-K -> bit-equal per image
    assert hip_ops.tune_plan(0, 1) == 0
    try:
        singles = []
        for i in range(B):
            cls_boxes, _, cls_keyps = engine.im_detect_all(model, ims[i], None)
            singles.append((cls_boxes, cls_keyps, ws.FetchBlob('rois').copy()))
        batch = engine.im_detect_all_batch(model, ims)
        rois = ws.FetchBlob('rois')
    finally:
        hip_ops.tune_plan(0, 0)
    for i in range(B):
        rb = rois[rois[:, 0] == i]
        np.testing.assert_array_equal(rb[:, 1:], singles[i][2][:, 1:], err_msg='proposals of image %d' % i)
        np.testing.assert_array_equal(batch[i][0][1], singles[i][0][1], err_msg='detections of image %d' % i)
        assert len(batch[i][2][1]) == len(singles[i][1][1])
        for a, b in zip(batch[i][2][1], singles[i][1][1]):
            np.testing.assert_array_equal(a, b, err_msg='keypoints of image %d' % i)
    # ---- pass 2: the planner's own plans (split-K where it pays): same results up to the fp32 summation order
    singles = []
    for i in range(B):
        cls_boxes, _, cls_keyps = engine.im_detect_all(model, ims[i], None)
        singles.append((cls_boxes, cls_keyps, ws.FetchBlob('rois').copy()))
    batch = engine.im_detect_all_batch(model, ims)
    assert ws.blobs['data'].t.shape[0] == B and len(batch) == B
    rois = ws.FetchBlob('rois')                         # images concatenated, col 0 = image index
    off = 0
    for i in range(B):
        r1 = singles[i][2]
        rb = rois[rois[:, 0] == i]
        assert np.all(rois[off:off + len(rb), 0] == i)              # image-major order
        off += len(rb)
        assert abs(len(rb) - len(r1)) <= 2, (i, len(rb), len(r1))
        assert _boxes_agree(rb[:, 1:], r1[:, 1:], 1e-2) > 0.97, i
        b1, k1 = singles[i][0][1], singles[i][1][1]
        bb, _, kb = batch[i]
        assert len(bb[1]) == len(b1) == 20 and len(kb[1]) == 20
        assert _boxes_agree(bb[1], b1, 2e-2) > 0.9, (i, bb[1][:3], b1[:3])
        # keypoints of the detections present in both: the decoded (x, y) agree (an arg-max over a 56 x 56 map upsampled to the box may
        # flip between near-equal cells under a 1e-6 logit difference: 95 % within one pixel)
        hit, diffs = 0, []
        for j, row in enumerate(bb[1]):
            d = np.abs(b1 - row[None]).max(axis=1)
            m = int(d.argmin())
            if d[m] < 2e-2:
                hit += 1
                diffs.append(np.abs(kb[1][j][:2] - k1[m][:2]).max(axis=0))
        assert hit >= 18
        assert (np.concatenate(diffs) < 1.0).mean() > 0.95
    # different images really are different results (the batch is not image 0 repeated)
    assert _boxes_agree(batch[0][0][1], batch[1][0][1], 1.0) < 0.5


def test_pipelined_engine_writes_the_same_detections_as_the_eager_loop(tmp_path):
    """core/test_engine.test_net on core/pipeline.ClipPipeline (uint8 upload, device pre-processing, several forwards in flight,
    hipGraph replay, completion-order read-back) must produce the detections.pkl of the reference's one-clip-at-a-time loop
    (lib/core/test_engine.py:124-204): bit-identical with one clip per forward -- the same kernels on a bit-identical `data` blob --
    and, with two clips per forward, the same detections up to the summation order of the larger conv grids."""
    import pickle
    from detectandtrack_amd.core import test_engine
    from detectandtrack_amd.core.config import cfg, cfg_from_cfg, assert_and_infer_cfg, reset_cfg
    from detectandtrack_amd import workspace
    T, H, W = 2, 96, 128
    rs = np.random.RandomState(5)
    roidb = [{'image': [rs.randint(0, 255, (H, W, 3)).astype(np.uint8) for _ in range(T)], 'height': H, 'width': W} for _ in range(7)]

    def run(depth, per, graph, out):
        c = fpn3d_kps_cfg('18', T=T, dtype='fp32', pre=300, post=100)     # (fp32: no exactly tied scores at the detection limit)
        c['TEST'].update(SCALES=(H,), MAX_SIZE=max(H, W), SCORE_THRESH=0.0, DETECTIONS_PER_IM=15)
        c['HIP'].update(PIPELINE_DEPTH=depth, IMS_PER_FORWARD=per, CLIP_GRAPH=graph)
        c['RNG_SEED'] = 3
        reset_cfg()
        cfg_from_cfg(c)
        assert_and_infer_cfg()
        workspace.ResetWorkspace()
        os.makedirs(out, exist_ok=True)
        res = test_engine.test_net(roidb, None, out)
        with open(os.path.join(out, 'detections.pkl'), 'rb') as f:
            disk = pickle.load(f)
        assert sorted(disk) == ['all_boxes', 'all_keyps', 'all_segms', 'cfg']
        return res, test_engine.test_net.last_stats
    eager, st0 = run(0, 1, False, str(tmp_path / 'eager'))
    assert st0 is None
    for depth, per, graph in ((3, 1, True), (2, 1, False)):
        got, st = run(depth, per, graph, str(tmp_path / ('p%d%d' % (depth, int(graph)))))
        assert st['clips'] == 7 and st['per_forward'] == 1 and st['upload_bytes_per_clip'] == T * H * W * 3
        for i in range(7):
            np.testing.assert_array_equal(got['all_boxes'][1][i], eager['all_boxes'][1][i])
            assert len(got['all_keyps'][1][i]) == len(eager['all_keyps'][1][i]) >= 15       # (>=: rows tied at the limit all stay)
            for a, b in zip(got['all_keyps'][1][i], eager['all_keyps'][1][i]):
                np.testing.assert_array_equal(a, b)
    got, st = run(2, 2, True, str(tmp_path / 'p22'))           # 7 clips = 3 forwards of two + a forward of one
    assert st['per_forward'] == 2
    for i in range(7):
        a, b = got['all_boxes'][1][i], eager['all_boxes'][1][i]
        assert a.shape[0] >= 15 and abs(a.shape[0] - b.shape[0]) <= 3 and a.shape[1] == 5
        assert _boxes_agree(a, b, 0.5) > 0.85, (i, a[:3], b[:3])


@pytest.mark.parametrize('per,graph', [(1, True), (2, True), (2, False)])
def test_pipelined_engine_with_the_frame_trunk_cache_writes_identical_detections(tmp_path, per, graph):
    """the per-frame trunk cache INSIDE the pipelined / hipGraph engine (core/pipeline.FrameTrunkCache).  A stride-1
    clip list of two videos (one clip per key frame, border frames replicated: lib/utils/video.py:149-201) through test_net with
    cfg.HIP.FRAME_TRUNK_CACHE: every video frame is uploaded and run through conv1 ... res2 exactly ONCE (also across the forwards in
    flight and across clips of one forward), the captured graphs start behind that prefix -- and detections.pkl is bit-identical to
    the same engine computing every clip whole (same kernels on bit-identical prefix outputs)."""
    from detectandtrack_amd.core import test_engine
    from detectandtrack_amd.core.config import cfg, cfg_from_cfg, assert_and_infer_cfg, reset_cfg
    from detectandtrack_amd import workspace
    T, H, W, n_frames = 4, 96, 128, 7
    rs = np.random.RandomState(9)
    roidb = []
    for v in range(2):
        video = [rs.randint(0, 255, (H, W, 3)).astype(np.uint8) for _ in range(n_frames)]
        for k in range(n_frames):
            ids = [min(max(k - T // 2 + j, 0), n_frames - 1) for j in range(T)]
            roidb.append({'image': [video[i] for i in ids], 'frame_ids': [('vid%d' % v, i) for i in ids], 'height': H, 'width': W})

    def run(cache, out):
        c = fpn3d_kps_cfg('18', T=T, dtype='fp32', pre=300, post=100)
        c['TEST'].update(SCALES=(H,), MAX_SIZE=max(H, W), SCORE_THRESH=0.0, DETECTIONS_PER_IM=15)
        c['HIP'].update(PIPELINE_DEPTH=3, IMS_PER_FORWARD=per, CLIP_GRAPH=graph, FRAME_TRUNK_CACHE=cache)
        c['RNG_SEED'] = 3
        reset_cfg()
        cfg_from_cfg(c)
        assert_and_infer_cfg()
        workspace.ResetWorkspace()
        os.makedirs(out, exist_ok=True)
        return test_engine.test_net(roidb, None, out), test_engine.test_net.last_stats
    plain, st0 = run(0, str(tmp_path / 'plain'))
    assert st0['frame_trunk_cache'] == 0 and st0['upload_bytes_per_clip'] == T * H * W * 3
    got, st = run(6 if per == 1 else 10, str(tmp_path / 'cached'))
    assert st['frame_trunk_cache'] > 0 and st['clips'] == len(roidb) and st['hip_graph'] == graph
    # padded tail groups repeat a clip whose frames are cached: requested counts the padding, computed never exceeds the real frames
    assert st['trunk_frames_computed'] == 2 * n_frames, st                    # every frame of both videos exactly once
    assert st['trunk_frames_requested'] >= len(roidb) * T
    assert st['upload_bytes_per_clip'] == 2 * n_frames * H * W * 3 / float(len(roidb))
    for i in range(len(roidb)):
        np.testing.assert_array_equal(got['all_boxes'][1][i], plain['all_boxes'][1][i], err_msg='clip %d' % i)
        assert len(got['all_keyps'][1][i]) == len(plain['all_keyps'][1][i]) >= 15
        for a, b in zip(got['all_keyps'][1][i], plain['all_keyps'][1][i]):
            np.testing.assert_array_equal(a, b)


def test_frame_trunk_cache_follows_a_mixed_resolution_clip_list(tmp_path):
    """PoseTrack videos differ in size.  A stride-1 clip list over THREE videos of two resolutions (A, B, A again) through
    the pipelined engine with cfg.HIP.FRAME_TRUNK_CACHE -- set smaller than one forward's frames on purpose: the pool of cached prefix outputs
    holds one geometry, so a frame of another size finishes what is in flight and starts an empty pool (before: an assert after the upload),
    and the capacity is raised to two forwards' worth of frames (before: a hard assert).  detections.pkl must be bit-identical to the same
    engine computing every clip whole; every frame still runs the prefix exactly once."""
    from detectandtrack_amd.core import test_engine
    from detectandtrack_amd.core.config import cfg, cfg_from_cfg, assert_and_infer_cfg, reset_cfg
    from detectandtrack_amd import workspace
    T, n_frames = 4, 5
    sizes = [(96, 128), (80, 144), (96, 128)]
    rs = np.random.RandomState(11)
    roidb = []
    for v, (H, W) in enumerate(sizes):
        video = [rs.randint(0, 255, (H, W, 3)).astype(np.uint8) for _ in range(n_frames)]
        for k in range(n_frames):
            ids = [min(max(k - T // 2 + j, 0), n_frames - 1) for j in range(T)]
            roidb.append({'image': [video[i] for i in ids], 'frame_ids': [('vid%d' % v, i) for i in ids], 'height': H, 'width': W})

    def run(cache, out):
        c = fpn3d_kps_cfg('18', T=T, dtype='fp32', pre=300, post=100)
        c['TEST'].update(SCALES=(96,), MAX_SIZE=160, SCORE_THRESH=0.0, DETECTIONS_PER_IM=15)
        c['HIP'].update(PIPELINE_DEPTH=3, IMS_PER_FORWARD=2, CLIP_GRAPH=True, FRAME_TRUNK_CACHE=cache)
        c['RNG_SEED'] = 3
        reset_cfg()
        cfg_from_cfg(c)
        assert_and_infer_cfg()
        workspace.ResetWorkspace()
        os.makedirs(out, exist_ok=True)
        return test_engine.test_net(roidb, None, out), test_engine.test_net.last_stats
    plain, st0 = run(0, str(tmp_path / 'plain'))
    assert st0['frame_trunk_cache'] == 0
    got, st = run(3, str(tmp_path / 'cached'))
    assert st['clips'] == len(roidb) and st['frame_trunk_cache'] >= 2 * 2 * T       # (capacity raised from 3 to two forwards' worth)
    assert st['trunk_resets'] == 2, st                                               # A -> B and B -> A
    assert st['trunk_frames_computed'] == len(sizes) * n_frames, st
    for i in range(len(roidb)):
        np.testing.assert_array_equal(got['all_boxes'][1][i], plain['all_boxes'][1][i], err_msg='clip %d' % i)
        assert len(got['all_keyps'][1][i]) == len(plain['all_keyps'][1][i]) >= 15
        for a, b in zip(got['all_keyps'][1][i], plain['all_keyps'][1][i]):
            np.testing.assert_array_equal(a, b)


def test_pipeline_graphs_survive_workspace_growth_and_a_second_geometry(monkeypatch):
    """ADVICE r3 (high + medium).  One pipeline slot holds a captured hipGraph per input geometry; the captured launches have the
    C-ABI context's scratch pointer baked in (proposal scratch, split-K partials).  (1) small geometry, then a LARGER one whose
    warm-up grows that scratch, then the small one again: the first graph must still replay correctly -- growth retires the outgrown
    buffer instead of freeing it (csrc/c_api.hip dat_ensure_ws).  (2) The exact-tie host fallback (`ClipPipeline._host_path`) reads
    `rois` / `cls_prob` / `bbox_pred` / the FPN blobs BY NAME: after the large geometry was captured last, a small forward that
    takes the fallback must read the tensors of the graph that was replayed, not the last captured one's.  (3) The per-slot graph
    cache is bounded (LRU)."""
    import torch
    from detectandtrack_amd.core import test as engine
    from detectandtrack_amd.core.pipeline import ClipPipeline
    from detectandtrack_amd.ops import hip_ops as ops
    T = 2
    c = fpn3d_kps_cfg('18', T=T, dtype='fp32', pre=300, post=100)
    c['TEST'].update(SCALES=(96,), MAX_SIZE=1000, SCORE_THRESH=0.0, DETECTIONS_PER_IM=15)
    model, ws, _ = build_product(c)
    rs = np.random.RandomState(7)
    small = [[rs.randint(0, 255, (96, 128, 3)).astype(np.uint8) for _ in range(T)] for _ in range(2)]
    # scale 96 / min side: a 96 x 320 frame keeps its size -> a 2.5x larger map AND more anchors than the small geometry
    large = [[rs.randint(0, 255, (96, 320, 3)).astype(np.uint8) for _ in range(T)] for _ in range(4)]
    ref_small = [engine.im_detect_all(model, clip, None) for clip in small]
    ref_large = engine.im_detect_all(model, large[0], None)

    pipe = ClipPipeline(model, ws, depth=1, graph=True, max_graphs=2)
    stream = pipe.slots[0].stream

    def info():
        with torch.cuda.stream(stream):
            return ops.ws_info()

    def run(clips):
        pipe.submit_frames(clips, tag='x')
        (_, out), = pipe.drain()
        return out

    def same(out, ref):
        np.testing.assert_array_equal(out[0][1], ref[0][1])
        assert len(out[2][1]) == len(ref[2][1])
        for a, b in zip(out[2][1], ref[2][1]):
            np.testing.assert_array_equal(a, b)
    same(run([small[0]])[0], ref_small[0])
    p0, n0, g0 = info()
    assert p0 and n0 > 0
    out = run(large)                                    # four clips of the larger geometry: the scratch must grow
    assert _boxes_agree(out[0][0][1], ref_large[0][1], 0.5) > 0.85      # (four clips per forward: other conv plans, not bit-equal)
    p1, n1, g1 = info()
    assert g1 > g0 and n1 > n0 and p1 != p0, 'the second geometry was meant to outgrow the scratch: %r -> %r' % ((p0, n0, g0), (p1, n1, g1))
    # (1) the first graph again, twice, with different clips: replayed launches still point at the retired buffer
    same(run([small[1]])[0], ref_small[1])
    same(run([small[0]])[0], ref_small[0])
    assert pipe.graphs_captured == 2 and pipe.graphs_evicted == 0
    # (2) force the host fallback for a SMALL forward while `ws.blobs` still names the large graph's tensors
    orig = engine.read_batch_results_from_device
    orig_one = engine.read_results_from_device
    monkeypatch.setattr(engine, 'read_batch_results_from_device', lambda *dev: [None] * len(orig(*dev)))
    monkeypatch.setattr(engine, 'read_results_from_device', lambda *dev: None)        # (the per-image re-run's reader)
    out = run([small[1]])
    assert pipe.host_path_images == 1 and pipe.rerun_images == 1      # (the forced overflow survives the device re-run too: host glue)
    a, b = out[0][0][1], ref_small[1][0][1]
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, atol=1e-4)
    assert len(out[0][2][1]) == len(ref_small[1][2][1])
    for x, y in zip(out[0][2][1], ref_small[1][2][1]):
        np.testing.assert_allclose(x[:2], y[:2], atol=1e-3)
    monkeypatch.setattr(engine, 'read_batch_results_from_device', orig)
    monkeypatch.setattr(engine, 'read_results_from_device', orig_one)
    # (3) a third geometry evicts the least recently used graph (the large one); the small one is still live
    third = [[rs.randint(0, 255, (96, 160, 3)).astype(np.uint8) for _ in range(T)]]
    run(third)
    assert pipe.graphs_captured == 3 and pipe.graphs_evicted == 1 and len(pipe.slots[0].graphs) == 2
    same(run([small[0]])[0], ref_small[0])
    assert pipe.graphs_captured == 3


def test_detections_tied_beyond_the_spare_rows_are_recomputed_on_the_device(monkeypatch):
    """an image whose detections tie exactly at the DETECTIONS_PER_IM cut beyond the device buffers' spare rows used
    to take the reference's HOST glue (a silent ~10x slower image).  The pipelined engine now re-runs the device glue + keypoint net with as
    many rows as the limit rule keeps.  Forced here by giving the first pass ZERO spare rows below the limit (out_cap < what the rule keeps):
    the overflow is detected from the device counts, the re-run returns the full result -- identical to the eager engine -- and the host
    path is never entered."""
    from detectandtrack_amd.core import test as engine
    from detectandtrack_amd.core.pipeline import ClipPipeline
    T = 2
    c = fpn3d_kps_cfg('18', T=T, dtype='fp32', pre=300, post=100)
    c['TEST'].update(SCALES=(96,), MAX_SIZE=1000, SCORE_THRESH=0.0, DETECTIONS_PER_IM=15)
    model, ws, _ = build_product(c)
    rs = np.random.RandomState(11)
    clips = [[rs.randint(0, 255, (96, 128, 3)).astype(np.uint8) for _ in range(T)] for _ in range(2)]
    ref = [engine.im_detect_all(model, clip, None) for clip in clips]
    orig = engine.enqueue_results_on_device
    calls = []

    def starved(model_, im_shape, im_scale, out_cap=None, image=None):
        calls.append((out_cap, image))
        return orig(model_, im_shape, im_scale, out_cap=9 if out_cap is None else out_cap, image=image)     # 9 rows for a limit of 15: overflow
    monkeypatch.setattr(engine, 'enqueue_results_on_device', starved)
    for graph in (False, True):
        pipe = ClipPipeline(model, ws, depth=1, graph=graph)
        pipe.submit_frames(clips, tag='x')
        (_, out), = pipe.drain()
        assert pipe.rerun_images == 2 and pipe.host_path_images == 0
        assert calls[-3][0] is None and [c[1] for c in calls[-2:]] == [0, 1] and all(c[0] >= 15 for c in calls[-2:])     # one re-run per image
        for o, r in zip(out, ref):
            assert o[0][1].shape == r[0][1].shape and _boxes_agree(o[0][1], r[0][1], 0.5) > 0.85     # (two clips per forward: other conv plans)
            assert len(o[2][1]) == len(r[2][1])


def test_pipeline_graphs_that_read_resident_inputs_in_place():
    """`ClipPipeline.submit(..., resident=True)` (round 4; what bench.py's `value` runs on): the caller's `data` blob stays where it is
    and the slot captures ONE graph per (geometry, buffer) that reads it in place -- no device-to-device copy into a private graph
    input.  Two resident blobs of one geometry on one slot: two graphs, each replay gives the detections of ITS buffer (bit-equal to
    the copying path and to the eager engine), and re-filling a buffer in place changes what its graph sees."""
    import torch
    from detectandtrack_amd.core import test as engine
    from detectandtrack_amd.core.pipeline import ClipPipeline
    from detectandtrack_amd.utils import blob as blob_utils
    T = 2
    c = fpn3d_kps_cfg('18', T=T, dtype='fp32', pre=300, post=100)
    c['TEST'].update(SCALES=(96,), MAX_SIZE=1000, SCORE_THRESH=0.0, DETECTIONS_PER_IM=15)
    model, ws, _ = build_product(c)
    rs = np.random.RandomState(9)
    clips = [[rs.randint(0, 255, (96, 128, 3)).astype(np.uint8) for _ in range(T)] for _ in range(3)]
    ref = [engine.im_detect_all(model, clip, None) for clip in clips]

    def blob_of(clip):
        u8 = torch.from_numpy(np.stack(clip)).cuda()
        data, _, im_info = blob_utils.frames_to_blob_on_device(u8, T)
        return data.clone(), im_info
    (a, info), (b, _), (c3, _) = blob_of(clips[0]), blob_of(clips[1]), blob_of(clips[2])
    pipe = ClipPipeline(model, ws, depth=1, graph=True)

    def run(data, resident):
        pipe.submit(data, info, (96, 128, 3), tag='r', resident=resident)
        (_, out), = pipe.drain()
        return out[0]

    def same(out, r):
        np.testing.assert_array_equal(out[0][1], r[0][1])
        for x, y in zip(out[2][1], r[2][1]):
            np.testing.assert_array_equal(x, y)
    same(run(a, True), ref[0])
    same(run(b, True), ref[1])
    assert pipe.graphs_captured == 2                    # one per resident buffer
    same(run(a, True), ref[0])
    same(run(b, True), ref[1])
    assert pipe.graphs_captured == 2
    a.copy_(c3)                                         # the caller re-fills its buffer in place: the graph reads the new clip
    torch.cuda.synchronize()
    same(run(a, True), ref[2])
    same(run(b, False), ref[1])                         # the copying path: a third graph with a private input
    assert pipe.graphs_captured == 3


@pytest.mark.gpu
def test_persistent_kernel_cu_share_does_not_change_results():
    """cfg.HIP.PERSISTENT_CU_SHARE (dat_conv3d_persistent_share, round 6): the pipelined engine gives the persistent HBM-bound conv kernels
    (res2's 3x3 convs, the P2 lateral, the weights-in-LDS 1x1 kernel) half of the CUs while several forwards are in flight.  Only the grid
    of those kernels changes: a layer's output is bit-identical at 100, 50 and 13 percent, one and two cout parts, with a residual."""
    from detectandtrack_amd.ops import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    cases = [(64, 64, (1, 3, 3), (0, 1, 1), False), (64, 256, (1, 1, 1), (0, 0, 0), True), (256, 64, (1, 1, 1), (0, 0, 0), False),
             (256, 1024, (1, 1, 1), (0, 0, 0), True)]
    T, H, W = 2, 96, 168
    for cin, cout, k, pads, with_res in cases:
        w = (torch.randn((cout, cin) + k, generator=g) * (2.0 / (cin * k[1] * k[2])) ** 0.5).cuda()
        layer = ops.ConvLayer(w, None, torch.zeros(cout).cuda(), stride=(1, 1), pads=pads, relu=True, dtype=ops.BF16)
        x = torch.randn((T, H, W, cin), generator=g).bfloat16().cuda()
        res = torch.randn((T, H, W, cout), generator=g).bfloat16().cuda() if with_res else None
        outs = []
        try:
            for pct in (100, 50, 13):
                ops.persistent_share(pct)
                outs.append(layer(x, T=T, residual=res).clone())
        finally:
            ops.persistent_share(100)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (cin, cout, k)
    with pytest.raises(Exception):
        ops.persistent_share(0)
