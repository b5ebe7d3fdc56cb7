"""Oracle (CPU restatement) vs golden vectors produced by the REAL reference code
(tests/golden/make_golden.py) and vs the reference's compiled Cython (oracle/_ref)."""
import numpy as np
import pytest

from oracle import anchors as oa
from oracle import boxes as ob
from oracle import nms as on
from oracle import proposals as op


def test_anchors_match_reference(golden):
    np.testing.assert_array_equal(oa.generate_anchors(16, (128, 256, 512), (0.5, 1, 2)), golden['anchors_s16_3x3'])
    np.testing.assert_array_equal(oa.generate_anchors(16., (64, 128, 256, 512), (0.5, 1, 2)), golden['anchors_c4_default'])
    np.testing.assert_array_equal(oa.generate_anchors(16., (64, 128, 256, 512), (0.5, 1, 2), time_dim=3),
                                  golden['anchors_c4_T3'])
    for lvl in range(2, 7):
        a = oa.generate_anchors(2. ** lvl, (32 * 2. ** (lvl - 2),), (0.5, 1, 2))
        np.testing.assert_array_equal(a, golden['anchors_fpn%d' % lvl])


def test_anchor_known_answer_table():
    # reference generate_anchors.py:16-39 lists the py-faster-rcnn table; the reference CODE
    # (anchor = [1,1,s,s]-1) yields that table shifted by exactly -1 in every coordinate.
    table = np.array([[-83, -39, 100, 56], [-175, -87, 192, 104], [-359, -183, 376, 200],
                      [-55, -55, 72, 72], [-119, -119, 136, 136], [-247, -247, 264, 264],
                      [-35, -79, 52, 96], [-79, -167, 96, 184], [-167, -343, 184, 360]], dtype=np.float64)
    np.testing.assert_array_equal(oa.generate_anchors(16, (128, 256, 512), (0.5, 1, 2)), table - 1)


def test_bbox_transform(golden):
    b, d = golden['bt_boxes'], golden['bt_deltas']
    np.testing.assert_array_equal(ob.bbox_transform(b.astype(np.float64), d, (1., 1., 1., 1.)), golden['bt_out_w1'])
    np.testing.assert_array_equal(ob.bbox_transform(b.astype(np.float64), d, (10., 10., 5., 5.)), golden['bt_out_w10'])
    tt = ob.bbox_transform(golden['tt_boxes'].astype(np.float64), golden['tt_deltas'], (10., 10., 5., 5.))
    np.testing.assert_array_equal(tt, golden['tt_out'])
    np.testing.assert_array_equal(ob.clip_tiled_boxes(tt.copy(), (256, 320)), golden['clip_out'])
    np.testing.assert_array_equal(ob.bbox_transform_inv(b, golden['inv_gt'], (10., 10., 5., 5.)), golden['inv_out'])


def test_bbox_transform_roundtrip():
    # reference tests/test_bbox_transform.py:41-53 checks transform(inv(src, dst)) == dst; with the
    # boxes.py that actually ships (x2 = ctr + 0.5*w, no "-1", utils/boxes.py:175-181) that identity
    # holds up to exactly +1 on x2/y2 — the golden vectors above pin the shipped behaviour.
    rs = np.random.RandomState(0)
    b = rs.uniform(0, 200, (50, 4)).astype(np.float32)
    b[:, 2:] += b[:, :2] + 5
    g = rs.uniform(0, 200, (50, 4)).astype(np.float32)
    g[:, 2:] += g[:, :2] + 5
    for w in ((1., 1., 1., 1.), (10., 10., 5., 5.)):
        d = ob.bbox_transform_inv(b, g, w)
        np.testing.assert_array_almost_equal(ob.bbox_transform(b, d, w), g + np.array([0, 0, 1, 1], np.float32), decimal=3)


def test_iou(golden):
    np.testing.assert_array_equal(ob.bbox_overlaps(golden['iou_a'], golden['iou_b']), golden['iou_out'])
    np.testing.assert_array_equal(ob.bbox_overlaps(golden['iou_ta'], golden['iou_tb']), golden['iou_tube_out'])


@pytest.mark.parametrize('key', ['nms_n300_t3', 'nms_n300_t5', 'nms_n1000_t7', 'nms_n1_t5', 'nms_n2_t5'])
def test_box_nms_bit_exact(golden, key):
    keep = on.nms(golden[key + '_dets'], float(key.split('_t')[1]) / 10.)
    np.testing.assert_array_equal(np.asarray(keep, dtype=np.int64), golden[key + '_keep'])


@pytest.mark.parametrize('key,thr', [('tnms_n200_T3_t5', 0.5), ('tnms_n120_T8_t7', 0.7), ('tnms_n1_T3_t5', 0.5)])
def test_tube_nms_bit_exact(golden, key, thr):
    keep = on.nms(golden[key + '_dets'], thr)
    np.testing.assert_array_equal(np.asarray(keep, dtype=np.int64), golden[key + '_keep'])


def test_nms_empty():
    assert list(on.nms(np.zeros((0, 5), np.float32), 0.5)) == []
    assert list(on.nms(np.zeros((0, 13), np.float32), 0.5)) == []


@pytest.mark.parametrize('name', ['gp_fpn3', 'gp_fpn2_min', 'gp_c4_T3'])
def test_generate_proposals(golden, name):
    stride, pre, post, thr, min_size = golden[name + '_cfg']
    rois, probs = op.generate_proposals(golden[name + '_scores'], golden[name + '_deltas'], golden[name + '_im_info'],
                                        golden[name + '_anchors'], 1. / stride, int(pre), int(post), float(thr),
                                        float(min_size))
    np.testing.assert_array_equal(rois, golden[name + '_rois'])
    np.testing.assert_array_equal(probs, golden[name + '_probs'])


def test_roi_to_batch(golden):
    np.testing.assert_array_equal(op.roi_to_batch_format(golden['r2b_in']), golden['r2b_out'])


def test_fpn_level_map_and_distribute(golden):
    np.testing.assert_array_equal(op.map_rois_to_fpn_levels(golden['lvl_rois'][:, 1:], 2, 5), golden['lvl_out'])
    rl = [golden['cd_rois%d' % i] for i in range(5)]
    sl = [golden['cd_scores%d' % i] for i in range(5)]
    rois = op.collect(rl, sl, 300)
    rois, per_level, restore = op.distribute(rois, 2, 5)
    np.testing.assert_array_equal(rois, golden['cd_out_rois'])
    for i in range(4):
        np.testing.assert_array_equal(per_level[i], golden['cd_out_fpn%d' % (i + 2)])
    np.testing.assert_array_equal(restore, golden['cd_out_restore'])


def test_oracle_vs_compiled_reference_live():
    """When oracle/_ref holds the compiled reference Cython, fuzz the oracle against it."""
    from oracle import build_ref
    mods = build_ref.load()
    if mods is None:
        pytest.skip('oracle/_ref not built')
    ref_nms, ref_bbox = mods
    rs = np.random.RandomState(11)
    for n in (0, 1, 5, 64, 257, 700):
        if n == 0:
            continue
        b = rs.uniform(0, 100, (n, 4)).astype(np.float32)
        b[:, 2:] = b[:, :2] + rs.uniform(1, 60, (n, 2)).astype(np.float32)
        d = np.hstack((b, rs.uniform(0, 1, (n, 1)).astype(np.float32)))
        for thr in (0.3, 0.5, 0.7):
            np.testing.assert_array_equal(on.nms_boxes(d, thr), ref_nms.nms(d, np.float32(thr)))
        q = rs.uniform(0, 100, (17, 4)).astype(np.float32)
        q[:, 2:] = q[:, :2] + rs.uniform(1, 60, (17, 2)).astype(np.float32)
        np.testing.assert_array_equal(ob.bbox_overlaps_2d(b, q), ref_bbox.bbox_overlaps(b, q))


def test_gpu_nms_restatement_known_answers():
    """`_nms` (lib/nms/nms_kernel.cu): strict > threshold, rows visited as given.  Hand-checkable cases: two 10x10 boxes shifted by
    5 px overlap 5*10 / (100 + 100 - 50) = 1/3: kept at thresh 1/3 exactly (strict), removed at any lower threshold, while the
    Cython CPU path (>=) removes it at 1/3 too."""
    from oracle import nms as onms
    b = np.array([[0, 0, 9, 9, 0.9], [5, 0, 14, 9, 0.8], [100, 100, 109, 109, 0.7]], np.float32)
    third = np.float32(50.0) / np.float32(150.0)
    assert onms.gpu_nms_presorted(b, third).tolist() == [0, 1, 2]
    assert onms.gpu_nms_presorted(b, np.nextafter(third, np.float32(0))).tolist() == [0, 2]
    assert onms.nms_boxes(b, third).tolist() == [0, 2]
    # order matters: rows are NOT re-sorted
    assert onms.gpu_nms_presorted(b[[1, 0, 2]], 0.3).tolist() == [0, 2]
    assert onms.gpu_nms_presorted(np.zeros((0, 5), np.float32), 0.5).tolist() == []


# ---- cv2.resize restatement (oracle/resize.py): known answers derived independently of the code under test -----------------
def _keys_kernel(x, A):
    """The published bicubic convolution kernel (Keys 1981) as exact rationals: W(x), support |x| < 2."""
    from fractions import Fraction
    x = abs(x)
    if x <= 1:
        return (A + 2) * x ** 3 - (A + 3) * x ** 2 + 1
    if x < 2:
        return A * x ** 3 - 5 * A * x ** 2 + 8 * A * x - 4 * A
    return Fraction(0)


def _exact_resize_1d(src, n_dst, kind):
    """Exact rational 1-D resample of integer samples with the pixel-centre mapping s = (d + 1/2) * n_src / n_dst - 1/2 and border
    replication: kind 'cubic' = sum_k src[clamp(floor(s) - 1 + k)] * W(s - (floor(s) - 1 + k)), A = -3/4; 'linear' = the two
    neighbours weighted by the fractional distance, coordinates clamped to the sample range."""
    from fractions import Fraction
    import math
    n = len(src)
    out = []
    for d in range(n_dst):
        s = (Fraction(d) + Fraction(1, 2)) * Fraction(n, n_dst) - Fraction(1, 2)
        if kind == 'linear':
            s = min(max(s, Fraction(0)), Fraction(n - 1))
        i0 = math.floor(s)
        if kind == 'linear':
            t = s - i0
            out.append(src[i0] * (1 - t) + src[min(i0 + 1, n - 1)] * t)
        else:
            acc = Fraction(0)
            for k in range(-1, 3):
                acc += src[min(max(i0 + k, 0), n - 1)] * _keys_kernel(s - (i0 + k), Fraction(-3, 4))
            out.append(acc)
    return out


def test_resize_oracle_identity_constant_and_ramp():
    from oracle import resize as R
    rs = np.random.RandomState(0)
    im = rs.randn(9, 13, 3).astype(np.float32)
    # same size: scale 1, every fractional offset 0 -> weights (0, 1, 0, 0) / (1, 0): an exact copy
    np.testing.assert_array_equal(R.resize_cubic(im, dsize=(13, 9)), im)
    np.testing.assert_array_equal(R.resize_linear(im, dsize=(13, 9)), im)
    # a constant map stays constant (the four cubic weights sum to 1 up to rounding)
    c = np.full((6, 5), 3.25, np.float32)
    assert np.abs(R.resize_cubic(c, dsize=(17, 11)) - 3.25).max() < 2e-6
    np.testing.assert_allclose(R.resize_linear(c, dsize=(17, 11)), 3.25, rtol=0, atol=5e-7)
    # a ramp: INTER_LINEAR reproduces it at the sample positions away from the border.  (The Keys kernel reproduces linear
    # functions only for A = -1/2; OpenCV's A = -3/4 does not -- what survives is the point symmetry about the centre.)
    ramp = np.tile(np.arange(16, dtype=np.float32) * 2.0 + 1.0, (4, 1))
    pos = (np.arange(23) + 0.5) * 16.0 / 23.0 - 0.5
    inner = (pos >= 1) & (pos <= 13.99)
    np.testing.assert_allclose(R.resize_linear(ramp, dsize=(23, 4))[0, inner], (pos * 2.0 + 1.0)[inner], atol=2e-5)
    out = R.resize_cubic(ramp, dsize=(23, 4))
    assert out[0, 11] == np.float32(16.0)
    np.testing.assert_allclose(out[0] + out[0, ::-1], 32.0, atol=1e-5)
    # weights at t = 1/2 (worked by hand from the kernel with A = -3/4): (-3/32, 19/32, 19/32, -3/32)
    np.testing.assert_array_equal(R.cubic_weights(np.float32(0.5)), np.array([-0.09375, 0.59375, 0.59375, -0.09375], np.float32))
    # [0 1 4 9] -> 7 samples: the centre sample sits at s = 1.5 -> -3/32*0 + 19/32*1 + 19/32*4 - 3/32*9 = 2.125 exactly
    assert R.resize_cubic(np.array([[0, 1, 4, 9]], np.float32), dsize=(7, 1))[0, 3] == np.float32(2.125)


@pytest.mark.parametrize('kind', ['cubic', 'linear'])
def test_resize_oracle_4x4_to_7x5_against_exact_rational_arithmetic(kind):
    """A 4 x 4 integer image resized to 7 wide x 5 high: every output pixel against the separable exact-rational evaluation of the
    published kernels (border replication included), to float32 rounding."""
    from fractions import Fraction
    from oracle import resize as R
    src = [[3, -1, 4, 1], [-5, 9, 2, -6], [5, 3, -5, 8], [9, -7, 9, 3]]
    rows = [_exact_resize_1d([Fraction(v) for v in r], 7, kind) for r in src]          # horizontal
    exact = [[None] * 7 for _ in range(5)]
    for x in range(7):
        col = _exact_resize_1d([rows[y][x] for y in range(4)], 5, kind)                # vertical
        for y in range(5):
            exact[y][x] = float(col[y])
    fn = R.resize_cubic if kind == 'cubic' else R.resize_linear
    got = fn(np.asarray(src, np.float32), dsize=(7, 5))
    assert got.shape == (5, 7)
    np.testing.assert_allclose(got, np.asarray(exact), rtol=0, atol=2e-5)   # float32 coordinates + float32 sums on values up to ~12


def test_resize_oracle_scale_given_uses_the_given_scale():
    """cv2.resize(im, None, fx=s, fy=s): output size round-half-even(n * s) and the sampling step 1/s (NOT src/dst) --
    lib/utils/blob.py:86-87 at the PoseTrack scale 1333/1280 on a 720-row frame: 750 rows, step 0.96024 (src/dst would be 0.96)."""
    from oracle import resize as R
    s = 1333.0 / 1280.0
    col = np.arange(720, dtype=np.float32).reshape(720, 1) * np.ones((1, 2), np.float32)
    out = R.resize_linear(col, fx=1.0, fy=s)
    assert out.shape == (750, 2)
    pos = np.clip((np.arange(750) + 0.5) / s - 0.5, 0, 719)
    np.testing.assert_allclose(out[:, 0], pos, atol=2e-4)
    assert abs(out[700, 0] - ((700.5) * 720.0 / 750.0 - 0.5)) > 0.1      # the dsize-derived step would land elsewhere
    assert R.resize_linear(np.zeros((5, 5), np.float32), fx=0.5, fy=0.5).shape == (2, 2)      # 2.5 -> 2 (half to even)
    assert R.resize_linear(np.zeros((7, 7), np.float32), fx=0.5, fy=0.5).shape == (4, 4)      # 3.5 -> 4


# ---- detection post-processing: the oracle's restatement against the REAL reference's functions --------------------------------
@pytest.fixture(scope='module')
def postproc():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_postproc.npz'))


@pytest.mark.parametrize('name', ['pp_boxes_k2', 'pp_boxes_k5', 'pp_tubes_k2', 'pp_nolimit'])
def test_box_results_oracle_matches_the_real_reference(postproc, name):
    """oracle/box_results.py vs core/test.py:750-806 of the reference run under py3 shims (tests/golden/make_golden.py): identical
    detections, order and scores; the decode (bbox_transform + clip) reproduces the reference's `pred` boxes."""
    from oracle import box_results as obr
    T, K, R, D, thr, nms_thr = postproc[name + '_cfg']
    T, K, D = int(T), int(K), int(D)
    boxes, scores, deltas = postproc[name + '_boxes'], postproc[name + '_scores'], postproc[name + '_deltas']
    rois = np.hstack((np.zeros((boxes.shape[0], 1), np.float32), boxes))
    sc, pred = obr.read_bbox_outputs(rois, scores, deltas, 1.0, (720, 1280, 3))
    np.testing.assert_array_equal(pred, postproc[name + '_pred'])
    out_s, out_b, cls_boxes = obr.box_results_with_nms_and_limit(sc, pred, K, thr, nms_thr, D)
    np.testing.assert_array_equal(out_s, postproc[name + '_out_scores'])
    np.testing.assert_array_equal(out_b, postproc[name + '_out_boxes'])
    assert [len(cls_boxes[j]) for j in range(1, K)] == postproc[name + '_out_counts'].tolist()


# ---- known answers for the two restated float operators that have no reference source in the tree ------------
# Legacy RoIAlign (Caffe2 modules/detectron @ b4e1588, call site lib/modeling/detector.py:240-245) and ConvTranspose k4 s2 p1
# (model_builder.py:848-856): every expected value below is worked by hand from the published definitions in exact dyadic
# rationals, so float32 reproduces it exactly.  tests/test_gpu_kernels.py runs the SAME cases through dat_roi_align / the sub-pixel
# deconv of the product.
def roi_align_known_answers():
    """[(name, feat (1,C,H,W), rois (R,5), pooled, scale, sampling, expected (R,C,P,P))]"""
    cases = []
    yy, xx = np.meshgrid(np.arange(6, dtype=np.float32), np.arange(8, dtype=np.float32), indexing='ij')
    ramp = (10.0 * xx + 100.0 * yy + 1.0)[None, None]                      # f(y, x) = 10 x + 100 y + 1: bilinear interpolation is exact on it
    # (1) a roi SMALLER THAN ONE PIXEL: roi_w = max(0.25, 1) = 1, roi_h = max(0.5, 1) = 1 -> bins of 0.5 from (x1, y1) = (2, 3);
    #     2 x 2 samples per bin at +0.125 / +0.375 -> bin centres x = 2.25 / 2.75, y = 3.25 / 3.75
    cases.append(('roi_smaller_than_a_pixel', ramp, np.array([[0, 2.0, 3.0, 2.25, 3.5]], np.float32), 2, 1.0, 2,
                  np.array([[[[10 * 2.25 + 325 + 1, 10 * 2.75 + 325 + 1], [10 * 2.25 + 375 + 1, 10 * 2.75 + 375 + 1]]]], np.float32)))
    # (2) the same roi given at 1/4 scale: corners are multiplied by spatial_scale BEFORE the max(., 1) clamp, no half-pixel shift
    cases.append(('spatial_scale_before_clamp', ramp, np.array([[0, 8.0, 12.0, 9.0, 14.0]], np.float32), 2, 0.25, 2,
                  cases[0][6]))
    # (3) samples left of the map: x1 = -3, roi_w = 4, one 2-wide bin per output column, samples at -2.5, -1.5 (both < -1: contribute 0)
    #     and -0.5 (in [-1, 0]: clamped onto column 0), 0.5; constant map 8 -> column 0 = 0, column 1 = 8; rows all inside
    const8 = np.full((1, 1, 4, 4), 8.0, np.float32)
    cases.append(('samples_left_of_minus_one_are_dropped', const8, np.array([[0, -3.0, 0.0, 1.0, 4.0]], np.float32), 2, 1.0, 2,
                  np.array([[[[0.0, 8.0], [0.0, 8.0]]]], np.float32)))
    # (4) a sample EXACTLY at -1 is kept (the test is y < -1), exactly at W is kept (x > W), beyond W dropped:
    #     x1 = -1.5, x2 = 0.5, pooled 1, sampling 2 -> samples -1.0 and 0.0 (both kept): 8;   y likewise from y1 = 3.0, y2 = 5.0 with H = 4:
    #     samples 3.5 (kept, clamped to row 3) and 4.5 (> H: dropped) -> half of the samples: 8 * 2 / 4 = 4
    cases.append(('samples_exactly_on_the_limits', const8, np.array([[0, -1.5, 3.0, 0.5, 5.0]], np.float32), 1, 1.0, 2,
                  np.array([[[[4.0]]]], np.float32)))
    #     ... and y1 = 2.0, y2 = 6.0: samples 3.0 and 5.0 -> 3.0 kept, 5.0 > 4 dropped; y1 = 2.0, y2 = 4.0 + 2.0 -> pick 4.0 exactly:
    #     y1 = 1.0, y2 = 5.0: bin 4, samples 2.0 and 4.0 (== H: kept, clamped) -> all four samples kept: 8
    cases.append(('a_sample_at_H_is_kept', const8, np.array([[0, -1.5, 1.0, 0.5, 5.0]], np.float32), 1, 1.0, 2,
                  np.array([[[[8.0]]]], np.float32)))
    # (5) clamping at the far border: a sample between the last column and W reads the LAST column (x_low >= W - 1 -> x = W - 1), it does
    #     not extrapolate: ramp f = 10 x + 1 on a 1 x 4 row; roi x in [3, 4], pooled 1, sampling 2 -> samples 3.25, 3.75 -> both 31 (not 33.5 / 38.5)
    row = (10.0 * np.arange(4, dtype=np.float32) + 1.0).reshape(1, 1, 1, 4)
    cases.append(('far_border_is_clamped_not_extrapolated', row, np.array([[0, 3.0, 0.0, 4.0, 1.0]], np.float32), 1, 1.0, 2,
                  np.array([[[[31.0]]]], np.float32)))
    # (6) plain bilinear value: [[1, 2], [3, 4]] sampled at (0.5, 0.5) = 2.5; sampling_ratio 0 = adaptive grid ceil(roi / pooled) = 1
    cases.append(('bilinear_centre_adaptive_grid', np.array([[[[1.0, 2.0], [3.0, 4.0]]]], np.float32),
                  np.array([[0, 0.0, 0.0, 1.0, 1.0]], np.float32), 1, 1.0, 0, np.array([[[[2.5]]]], np.float32)))
    #     adaptive grid on a 3-wide roi, pooled 1: ceil(3) = 3 samples per axis at 0.5, 1.5, 2.5 of the ramp f = 10 x + 100 y + 1 -> mean at (1.5, 1.5)
    cases.append(('adaptive_grid_three_samples', ramp, np.array([[0, 0.0, 0.0, 3.0, 3.0]], np.float32), 1, 1.0, 0,
                  np.array([[[[10 * 1.5 + 100 * 1.5 + 1]]]], np.float32)))
    # (7) two channels, two rois, the second roi on a second image of the batch
    two = np.concatenate([ramp, 2.0 * ramp], axis=1)
    batch = np.concatenate([two, -two], axis=0)
    cases.append(('channels_and_batch_index', batch, np.array([[0, 2.0, 3.0, 2.25, 3.5], [1, 2.0, 3.0, 2.25, 3.5]], np.float32), 2, 1.0, 2,
                  np.concatenate([np.concatenate([cases[0][6], 2 * cases[0][6]], axis=1),
                                  -np.concatenate([cases[0][6], 2 * cases[0][6]], axis=1)], axis=0)))
    return cases


@pytest.mark.parametrize('case', roi_align_known_answers(), ids=lambda c: c[0])
def test_roi_align_oracle_known_answers(case):
    from oracle.roi_align import roi_align_2d
    name, feat, rois, pooled, scale, sampling, exp = case
    np.testing.assert_array_equal(roi_align_2d(feat, rois, pooled, scale, sampling), exp)


def conv_transpose_k4s2p1_known_answer():
    """x (1, 2, 2, 2), w (2, 1, 4, 4) [Cin, Cout, kh, kw], bias -> out (1, 1, 4, 4), by the definition out[o] = sum_i x[i] w[o - 2 i + 1]
    (k = o - s i + p in [0, 3]; out = (in - 1) s - 2 p + k = 2 in).  1-D by hand for x = [1, 2], w = [1, 10, 100, 1000]:
    o=0: i=0,k=1 -> 10;  o=1: (i=0,k=2) 100 + (i=1,k=0) 2 = 102;  o=2: (i=0,k=3) 1000 + (i=1,k=1) 20 = 1020;  o=3: i=1,k=2 -> 200.
    Channel 0 uses the separable kernel outer(w, w) on x0 = outer([1, 2], [1, 2]) -> outer(v, v) with v = [10, 102, 1020, 200];
    channel 1 uses the one-hot kernel e(kh=2, kw=1) on x1 = [[1, 2], [3, 4]]: out[2 i + 1, 2 j] = x1[i, j] (o = 2 i + k - 1)."""
    w1 = np.array([1.0, 10.0, 100.0, 1000.0], np.float32)
    v = np.array([10.0, 102.0, 1020.0, 200.0], np.float32)
    x = np.zeros((1, 2, 2, 2), np.float32)
    x[0, 0] = np.outer([1.0, 2.0], [1.0, 2.0])
    x[0, 1] = [[1.0, 2.0], [3.0, 4.0]]
    w = np.zeros((2, 1, 4, 4), np.float32)
    w[0, 0] = np.outer(w1, w1)
    w[1, 0, 2, 1] = 1.0
    exp = np.outer(v, v).astype(np.float32)
    exp[1, 0] += 1.0
    exp[1, 2] += 2.0
    exp[3, 0] += 3.0
    exp[3, 2] += 4.0
    bias = np.array([0.5], np.float32)
    return x, w, bias, (exp + 0.5)[None, None]


def test_conv_transpose_oracle_known_answer():
    """What oracle/net3d.py::kps_outputs_2d uses for `kps_score_lowres` (F.conv_transpose2d, stride 2, padding 1) against the
    hand-worked definition -- all values are small integers + 0.5: exact in fp32."""
    import torch
    import torch.nn.functional as F
    x, w, b, exp = conv_transpose_k4s2p1_known_answer()
    got = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=2, padding=1).numpy()
    assert got.shape == (1, 1, 4, 4)
    np.testing.assert_array_equal(got, exp)


def test_bilinear_upsampling_kernel_known_answer():
    """detector.py:348-380 (BilinearInterpolation: a fixed ConvTranspose, kernel 2 up, stride up, pad up / 2, weights from
    upsample_filt) at UP_SCALE 2: the 1-D taps are (0.25, 0.75, 0.75, 0.25) and only the diagonal (k -> k) is non-zero; a constant
    map stays constant in the interior, a ramp of slope 1 gets slope 1/2."""
    import torch
    import torch.nn.functional as F
    from oracle.net3d import bilinear_kernel
    k = bilinear_kernel(3, 2)
    assert k.shape == (3, 3, 4, 4)
    t = np.array([0.25, 0.75, 0.75, 0.25], np.float32)
    for i in range(3):
        for j in range(3):
            np.testing.assert_array_equal(k[i, j], np.outer(t, t) if i == j else np.zeros((4, 4), np.float32))
    low = torch.arange(5, dtype=torch.float32).view(1, 1, 1, 5).repeat(1, 3, 5, 1)          # value = column index
    up = F.conv_transpose2d(low, torch.from_numpy(k), None, stride=2, padding=1).numpy()
    assert up.shape == (1, 3, 10, 10)
    np.testing.assert_array_equal(up[0, 1, 4, 1:9], np.array([0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3.25, 3.75], np.float32))
