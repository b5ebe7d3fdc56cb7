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
