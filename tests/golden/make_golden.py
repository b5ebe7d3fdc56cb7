#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REAL reference Python
(/root/reference/lib, py2 code) under small py3 shims, plus the reference's
Cython NMS/IoU compiled into oracle/_ref (oracle/build_ref.py).

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py            (everything; or one generator: --only-cfg | --only-roi-data | --only-lr | --only-blob |
                                                   --only-decode | --only-tracker | --only-builders | --only-weights | --only-clips | --only-postproc | --only-dataset)
Writes tests/golden/: reference_host.npz (anchors, transforms, IoU / NMS, GenerateProposals, RoIToBatchFormat, level mapping, collect /
distribute, inflation), reference_roi_data.npz (training labels, boxes and tubes), reference_lr_policy.npz (schedules + the momentum
correction rule), reference_postproc.npz, reference_posetrack_annorect.json, reference_blob.npz, reference_decode.npz,
reference_tracker.json, reference_cfg_defaults.json / reference_cfg_files.json, reference_builder_nets.json.gz (the graphs the
reference's builder functions emit) + reference_heatmap_outputs.json (the keypoint output function on a 3D head, both deconv variants), reference_weights_load.npz (checkpoint loading), reference_clips.json (clip assembly), synthetic_posetrack.json + reference_json_dataset.npz / .json (the dataset layer: JsonDataset.get_roidb and get_clip's tube ground truth on a synthetic COCO-format file).  The shims do not change any arithmetic:
  * removed NumPy aliases (np.float/np.int), py2 builtins (basestring, unicode),
    cPickle -> pickle, bytes config defaults decoded to str;
  * caffe2 / cv2 / pycocotools are replaced by inert stub modules so that pure
    NumPy functions living in files that import them can be imported.
"""
import builtins
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        full = self.__name__ + '.' + name
        m = _Stub(full)
        sys.modules[full] = m
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return _Stub(self.__name__ + '()')

    def __enter__(self):            # (`with core.DeviceScope(...)`, `with core.NameScope(...)`)
        return self

    def __exit__(self, *exc):
        return False


def _install_shims():
    np.float = float
    np.int = int
    builtins.basestring = str
    builtins.unicode = str
    sys.modules['cPickle'] = pickle
    for root in ('caffe2', 'cv2', 'pycocotools', 'h5py'):
        sys.modules[root] = _Stub(root)
    for sub in ('caffe2.python', 'caffe2.python.core', 'caffe2.python.workspace', 'caffe2.python.scope',
                'caffe2.python.cnn', 'caffe2.python.muji', 'caffe2.python.utils', 'caffe2.proto',
                'caffe2.proto.caffe2_pb2', 'pycocotools.mask', 'pycocotools.coco', 'pycocotools.cocoeval',
                'caffe2.python.modeling', 'caffe2.python.modeling.parameter_info', 'caffe2.python.memonger',
                'caffe2.python.dyndep'):
        parent, _, leaf = sub.rpartition('.')
        getattr(sys.modules[parent], leaf)
    from oracle import build_ref
    build_ref.build()
    nms_mod, bbox_mod = build_ref.load()
    sys.path.insert(0, '/root/reference/lib')
    import utils  # noqa  (reference package)
    sys.modules['utils.cython_nms'] = nms_mod
    sys.modules['utils.cython_bbox'] = bbox_mod
    from core.config import cfg

    def fix(d):
        for k, v in d.items():
            if isinstance(v, bytes):
                d[k] = v.decode()
            elif isinstance(v, dict):
                fix(v)
    fix(cfg)
    return cfg


class _Blob(object):
    def __init__(self, data=None):
        self.data = data
        self.shape = None if data is None else data.shape

    def reshape(self, shape):
        self.data = np.zeros(shape, dtype=np.float32)
        self.shape = tuple(shape)

    def init(self, shape, dtype_code):  # utils/blob.py py_op_copy_blob int32 path
        self.data = np.zeros(shape, dtype=np.int32)
        self.shape = tuple(shape)


def main():
    cfg = _install_shims()
    golden_cfg_defaults(cfg)
    golden_cfg_files(cfg)
    from modeling.generate_anchors import generate_anchors
    import utils.boxes as box_utils
    from core.nms_wrapper import nms
    from ops.generate_proposals import GenerateProposalsOp
    from ops.roi_blob_transforms import RoIToBatchFormatOp
    rs = np.random.RandomState(3)
    out = {}

    # ---- anchors -------------------------------------------------------------------
    out['anchors_s16_3x3'] = generate_anchors(16, (128, 256, 512), (0.5, 1, 2))
    out['anchors_c4_default'] = generate_anchors(16., cfg.RPN.SIZES, cfg.RPN.ASPECT_RATIOS, time_dim=1)
    out['anchors_c4_T3'] = generate_anchors(16., cfg.RPN.SIZES, cfg.RPN.ASPECT_RATIOS, time_dim=3)
    for lvl in range(2, 7):
        out['anchors_fpn%d' % lvl] = generate_anchors(2. ** lvl, (32 * 2. ** (lvl - 2),), (0.5, 1, 2), time_dim=1)

    # ---- bbox transforms ---------------------------------------------------------------
    def rand_boxes(n, T=1, W=320., H=256.):
        b = np.zeros((n, 4 * T), dtype=np.float32)
        for t in range(T):
            x1 = rs.uniform(0, W - 20, n)
            y1 = rs.uniform(0, H - 20, n)
            b[:, 4 * t + 0] = x1
            b[:, 4 * t + 1] = y1
            b[:, 4 * t + 2] = x1 + rs.uniform(4, 120, n)
            b[:, 4 * t + 3] = y1 + rs.uniform(4, 120, n)
        return b
    boxes = rand_boxes(64)
    deltas = (rs.randn(64, 4) * 0.5).astype(np.float32)
    out['bt_boxes'], out['bt_deltas'] = boxes, deltas
    out['bt_out_w1'] = box_utils.bbox_transform(boxes.astype(np.float64), deltas, (1., 1., 1., 1.))
    out['bt_out_w10'] = box_utils.bbox_transform(boxes.astype(np.float64), deltas, (10., 10., 5., 5.))
    tb = rand_boxes(32, T=3)
    td = (rs.randn(32, 2 * 3 * 4) * 0.5).astype(np.float32)  # 2 classes x 3 frames
    out['tt_boxes'], out['tt_deltas'] = tb, td
    out['tt_out'] = box_utils.bbox_transform(tb.astype(np.float64), td, (10., 10., 5., 5.))
    out['clip_out'] = box_utils.clip_tiled_boxes(out['tt_out'].copy(), (256, 320))
    gt = rand_boxes(64)
    out['inv_gt'] = gt
    out['inv_out'] = box_utils.bbox_transform_inv(boxes, gt, (10., 10., 5., 5.))

    # ---- IoU + NMS (reference Cython, compiled) ----------------------------------------------
    a = rand_boxes(50)
    b = rand_boxes(70)
    out['iou_a'], out['iou_b'] = a, b
    out['iou_out'] = box_utils.bbox_overlaps(a, b)
    ta, tb2 = rand_boxes(20, T=3), rand_boxes(30, T=3)
    out['iou_ta'], out['iou_tb'] = ta, tb2
    out['iou_tube_out'] = box_utils.bbox_overlaps(ta, tb2)
    for n, thr in ((300, 0.3), (300, 0.5), (1000, 0.7), (1, 0.5), (2, 0.5)):
        d = np.hstack((rand_boxes(n, W=200., H=160.), rs.uniform(0, 1, (n, 1)).astype(np.float32))).astype(np.float32)
        key = 'nms_n%d_t%d' % (n, int(thr * 10))
        out[key + '_dets'] = d
        out[key + '_keep'] = np.asarray(nms(d, thr), dtype=np.int64)
    for n, T, thr in ((200, 3, 0.5), (120, 8, 0.7), (1, 3, 0.5)):
        d = rand_boxes(n, T=1, W=200., H=160.)
        jit = [d + rs.uniform(-6, 6, d.shape).astype(np.float32) for _ in range(T)]
        d = np.hstack(jit + [rs.uniform(0, 1, (n, 1)).astype(np.float32)]).astype(np.float32)
        key = 'tnms_n%d_T%d_t%d' % (n, T, int(thr * 10))
        out[key + '_dets'] = d
        out[key + '_keep'] = np.asarray(nms(d, thr), dtype=np.int64)

    # ---- GenerateProposalsOp (the real op class) -------------------------------------------------
    def run_gp(name, A_anchors, H, W, T, stride, pre, post, thr, min_size, im_hw):
        cfg.TEST.RPN_PRE_NMS_TOP_N = pre
        cfg.TEST.RPN_POST_NMS_TOP_N = post
        cfg.TEST.RPN_NMS_THRESH = thr
        cfg.TEST.RPN_MIN_SIZE = min_size
        A = A_anchors.shape[0]
        scores = rs.uniform(0, 1, (1, A, H, W)).astype(np.float32)
        deltas = (rs.randn(1, 4 * A * T, H, W) * 0.3).astype(np.float32)
        im_info = np.array([[im_hw[0], im_hw[1], 1.25]], dtype=np.float32)
        op = GenerateProposalsOp(A_anchors, 1. / stride, False)
        outs = [_Blob(), _Blob()]
        op.forward([_Blob(scores), _Blob(deltas), _Blob(im_info)], outs)
        out[name + '_scores'], out[name + '_deltas'], out[name + '_im_info'] = scores, deltas, im_info
        out[name + '_anchors'] = A_anchors
        out[name + '_cfg'] = np.array([stride, pre, post, thr, min_size], dtype=np.float64)
        out[name + '_rois'], out[name + '_probs'] = outs[0].data, outs[1].data
    run_gp('gp_fpn3', out['anchors_fpn3'], 32, 40, 1, 8., 1000, 300, 0.7, 0, (256, 320))
    run_gp('gp_fpn2_min', out['anchors_fpn2'], 24, 28, 1, 4., 500, 200, 0.7, 16, (96, 112))
    run_gp('gp_c4_T3', out['anchors_c4_T3'], 16, 20, 3, 16., 600, 150, 0.7, 0, (256, 320))

    # ---- RoIToBatchFormat (real op) ------------------------------------------------------------------
    tr = np.hstack((np.zeros((5, 1), np.float32), rand_boxes(5, T=3)))
    ob = [_Blob()]
    RoIToBatchFormatOp().forward([_Blob(tr)], ob)
    out['r2b_in'], out['r2b_out'] = tr, ob[0].data

    # ---- FPN level mapping + collect/distribute (real functions; caffe2 stubbed) ----------------------
    import modeling.FPN as fpn
    import ops.collect_and_distribute_fpn_rpn_proposals as cd
    rois = np.hstack((np.zeros((400, 1), np.float32), rand_boxes(400, W=1300., H=740.)))
    rois[:, 3] = rois[:, 1] + rs.uniform(4, 700, 400)
    rois[:, 4] = rois[:, 2] + rs.uniform(4, 600, 400)
    out['lvl_rois'] = rois
    out['lvl_out'] = fpn.map_rois_to_fpn_levels(rois[:, 1:], 2, 5)
    cfg.TEST.RPN_POST_NMS_TOP_N = 300
    cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.RPN_MIN_LEVEL = 6, 2
    rl = [np.hstack((np.zeros((n, 1), np.float32), rand_boxes(n, W=1300., H=740.))) for n in (120, 100, 80, 40, 10)]
    sl = [rs.uniform(0, 1, (r.shape[0], 1)).astype(np.float32) for r in rl]
    for i in range(5):
        out['cd_rois%d' % i], out['cd_scores%d' % i] = rl[i], sl[i]
    collected = cd.collect([_Blob(r) for r in rl] + [_Blob(s) for s in sl], False)
    outs = [_Blob() for _ in range(6)]
    # blob_utils.py_op_copy_blob is a caffe2-stubbed import here; replicate its int32 copy by hand
    import utils.blob as blob_utils
    blob_utils.py_op_copy_blob = lambda arr, blob: setattr(blob, 'data', np.array(arr))
    cd.blob_utils = blob_utils
    cd.distribute(collected, None, outs, False)
    out['cd_out_rois'] = outs[0].data
    for i in range(4):
        out['cd_out_fpn%d' % (i + 2)] = outs[1 + i].data
    out['cd_out_restore'] = np.asarray(outs[5].data, dtype=np.int32)

    # ---- weight inflation (utils/net.py:95-161) ---------------------------------------------------------
    import utils.net as net_utils
    w2d = rs.randn(8, 4, 3, 3).astype(np.float32)
    for mode in ('mean-repeat', 'repeat', 'center-only'):
        cfg.VIDEO.WEIGHTS_INFLATE_MODE = mode
        out['inflate_' + mode.replace('-', '_')] = net_utils.inflate_weights(
            w2d, np.zeros((8, 4, 3, 3, 3), np.float32), 'x_w', {'x_w': w2d})
    out['inflate_src'] = w2d

    np.savez_compressed(os.path.join(HERE, 'reference_host.npz'), **out)
    print('wrote', os.path.join(HERE, 'reference_host.npz'), len(out), 'arrays')
    golden_roi_data(cfg)
    golden_lr_policy(cfg)


LR_CASES = [
    # LR_POLICY, BASE_LR, GAMMA, STEP_SIZE, STEPS, LRS, MAX_ITER, WARM_UP_ITERS, WARM_UP_FACTOR, WARM_UP_METHOD
    ('steps_with_decay', 0.02, 0.1, 30000, [0, 60, 80], [], 90, 20, 1.0 / 3.0, 'linear'),
    ('steps_with_decay', 0.002, 0.5, 30000, [0, 10, 35, 70], [], 100, 0, 1.0 / 3.0, 'linear'),
    ('steps_with_lrs', 0.02, 0.1, 30000, [0, 40, 75], [0.02, 0.004, 0.0001], 90, 15, 0.1, 'constant'),
    ('step', 0.01, 0.3, 25, [0], [], 100, 8, 0.25, 'linear'),
]


def golden_lr_policy(cfg):
    """SOLVER schedules of the REAL reference lib/utils/lr_policy.py for iterations 0 .. MAX_ITER + 9."""
    import utils.lr_policy as lr_policy
    out = {}
    for i, (pol, base, gamma, step_size, steps, lrs, max_iter, wi, wf, wm) in enumerate(LR_CASES):
        so = cfg.SOLVER
        so.LR_POLICY, so.BASE_LR, so.GAMMA, so.STEP_SIZE, so.STEPS, so.LRS = pol, base, gamma, step_size, list(steps), list(lrs)
        so.MAX_ITER, so.WARM_UP_ITERS, so.WARM_UP_FACTOR, so.WARM_UP_METHOD = max_iter, wi, wf, wm
        out['lr_case%d' % i] = np.array([lr_policy.get_lr_at_iter(it) for it in range(max_iter + 10)], dtype=np.float32)
    # the momentum correction at a learning-rate change: lib/modeling/detector.py:606-616 _SetNewLr ITSELF (unbound, on a dummy whose
    # _CorrectMomentum records the factor; the Caffe2 FeedBlob it makes first is a stub)
    import queue
    sys.modules['Queue'] = queue
    sys.modules['caffe2.python.cnn'].CNNModelHelper = type('CNNModelHelper', (object,), {})
    import modeling.detector as rdet
    pairs = [(0.0, 0.02), (0.02, 0.002), (0.002, 0.02), (0.02, 0.021), (0.02, 0.0225), (0.02, 0.018), (0.02, 0.0181), (1e-8, 0.01),
             (0.01, 0.0), (0.006666667, 0.007333333), (5e-5, 5e-4), (0.02, 0.02)]
    rows = []
    for mode in (True, False):
        cfg.SOLVER.SCALE_MOMENTUM = mode
        for cur, new in pairs:
            calls = []

            class Dummy(object):
                def _CorrectMomentum(self, c):
                    calls.append(c)
            with np.errstate(divide='ignore', invalid='ignore'):
                rdet.DetectionModelHelper._SetNewLr(Dummy(), np.float32(cur), np.float32(new))
            rows.append([float(mode), float(np.float32(cur)), float(np.float32(new)), float(len(calls)), float(calls[0]) if calls else 0.0])
    cfg.SOLVER.SCALE_MOMENTUM = True
    out['momentum_correction'] = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'reference_lr_policy.npz'), **out)
    print('wrote reference_lr_policy.npz', len(out), 'schedules')


def golden_roi_data(cfg):
    """Training label generation of the REAL reference (lib/roi_data/rpn.py, fast_rcnn.py, keypoint_rcnn.py,
    datasets/json_dataset.py:_merge_proposal_boxes_into_roidb, utils/keypoints.py:keypoints_to_heatmap_labels) on a seeded
    synthetic roidb record -> tests/golden/reference_roi_data.npz.  numpy.random is seeded right before each reference call; the
    host restatement (detectandtrack_amd/roi_data) consumes the global generator in the same order."""
    import numpy.random as npr
    import scipy.sparse
    import roi_data.rpn as rrpn
    import roi_data.fast_rcnn as rfr
    import datasets.json_dataset as jd
    cfg.FPN.FPN_ON = True
    cfg.FPN.MULTILEVEL_RPN = True
    cfg.FPN.MULTILEVEL_ROIS = True
    cfg.MODEL.KEYPOINTS_ON = True
    cfg.MODEL.NUM_CLASSES = 2
    cfg.KRCNN.NUM_KEYPOINTS = 17
    cfg.KRCNN.HEATMAP_SIZE = 56
    cfg.TRAIN.MAX_SIZE = 333
    cfg.TRAIN.BATCH_SIZE_PER_IM = 64
    H, W, n = 256, 320, 5
    rs = np.random.RandomState(21)
    bw, bh = rs.uniform(0.15, 0.5, n) * W, rs.uniform(0.25, 0.8, n) * H
    x1, y1 = rs.uniform(0, 1, n) * (W - bw - 1), rs.uniform(0, 1, n) * (H - bh - 1)
    boxes = np.stack([x1, y1, x1 + bw, y1 + bh], axis=1).astype(np.float32)
    kps = np.zeros((n, 3, 17), dtype=np.int32)
    kps[:, 0, :] = (x1[:, None] + rs.uniform(-0.1, 1.1, (n, 17)) * bw[:, None]).astype(np.int32)
    kps[:, 1, :] = (y1[:, None] + rs.uniform(-0.1, 1.1, (n, 17)) * bh[:, None]).astype(np.int32)
    kps[:, 2, :] = rs.randint(0, 3, (n, 17))
    ov = np.zeros((n, 2), dtype=np.float32)
    ov[:, 1] = 1.0
    out = dict(rd_boxes=boxes, rd_kps=kps, rd_hw=np.array([H, W]))
    # ---- RPN labels ------------------------------------------------------------------------------------------------------
    foas = []
    for lvl in range(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.RPN_MAX_LEVEL + 1):
        foas.append(rrpn._get_field_of_anchors(2. ** lvl, (cfg.FPN.RPN_ANCHOR_START_SIZE * 2. ** (lvl - cfg.FPN.RPN_MIN_LEVEL),),
                                               cfg.FPN.RPN_ASPECT_RATIOS, 1))
    all_anchors = np.concatenate([f.field_of_anchors for f in foas])
    npr.seed(77)
    blobs = rrpn._get_rpn_blobs(float(H), float(W), foas, all_anchors, boxes, np.full((n, 1), True))
    for i, b in enumerate(blobs):
        for k, v in b.items():
            if 'vis' not in k:
                out['rd_%s_fpn%d' % (k, i + 2)] = v
    # ---- proposals -> roidb -> sampled rois + keypoint targets ---------------------------------------------------------------
    props = np.clip(boxes[rs.randint(0, n, 300)] + rs.randn(300, 4) * 18, 0, [W - 1, H - 1, W - 1, H - 1]).astype(np.float32)
    props = props[(props[:, 2] > props[:, 0] + 2) & (props[:, 3] > props[:, 1] + 2)]
    out['rd_props'] = props
    entry = dict(boxes=boxes.copy(), gt_classes=np.ones((n,), np.int32), is_crowd=np.zeros((n,), np.bool_),
                 gt_overlaps=scipy.sparse.csr_matrix(ov), box_to_gt_ind_map=np.arange(n, dtype=np.int32),
                 gt_keypoints=kps.copy(), seg_areas=np.zeros((n,), np.float32), segms=[[] for _ in range(n)],
                 height=H, width=W)
    roidb = [entry]
    jd._merge_proposal_boxes_into_roidb(roidb, [props])
    jd._add_class_assignments(roidb)
    out['rd_merged_max_overlaps'] = roidb[0]['max_overlaps']
    out['rd_merged_b2g'] = roidb[0]['box_to_gt_ind_map']
    npr.seed(78)
    np.random.seed(78)
    sb = rfr._sample_rois(roidb[0], 1.0, 0)
    for k, v in sb.items():
        out['rd_s_' + k] = np.asarray(v)
    # ---- the same chain on TUBES (T = 3: boxes n x 4T, keypoints n x 3 x 17T, tube proposals): tube IoU in the merge, 4T-wide class-specific
    #      targets (fast_rcnn.py:206-229), the first frame's box against all T x 17 keypoints in the visibility test and per-frame heatmap
    #      cells (keypoint_rcnn.py:62-99) ------------------------------------------------------------------------------------------
    T, n = 3, 4
    cfg.MODEL.VIDEO_ON, cfg.VIDEO.NUM_FRAMES, cfg.VIDEO.NUM_FRAMES_MID = True, T, T
    rs = np.random.RandomState(22)
    bw, bh = rs.uniform(0.15, 0.5, n) * W, rs.uniform(0.25, 0.8, n) * H
    x1, y1 = rs.uniform(0, 1, n) * (W - bw - 1), rs.uniform(0, 1, n) * (H - bh - 1)
    tubes = np.zeros((n, 4 * T), np.float32)
    tkps = np.zeros((n, 3, 17 * T), np.int32)
    for t in range(T):
        dx = 3.0 * t
        tubes[:, 4 * t:4 * t + 4] = np.stack([x1 + dx, y1, x1 + bw + dx, y1 + bh], axis=1)
        tkps[:, 0, 17 * t:17 * t + 17] = (x1[:, None] + dx + rs.uniform(-0.1, 1.1, (n, 17)) * bw[:, None]).astype(np.int32)
        tkps[:, 1, 17 * t:17 * t + 17] = (y1[:, None] + rs.uniform(-0.1, 1.1, (n, 17)) * bh[:, None]).astype(np.int32)
        tkps[:, 2, 17 * t:17 * t + 17] = rs.randint(0, 3, (n, 17))
    tov = np.zeros((n, 2), np.float32)
    tov[:, 1] = 1.0
    tprops = np.clip(tubes[rs.randint(0, n, 200)] + np.tile(rs.randn(200, 4) * 15, (1, T)), 0, np.tile([W - 1, H - 1, W - 1, H - 1], T)).astype(np.float32)
    ok = np.all([(tprops[:, 4 * t + 2] > tprops[:, 4 * t] + 2) & (tprops[:, 4 * t + 3] > tprops[:, 4 * t + 1] + 2) for t in range(T)], axis=0)
    tprops = tprops[ok]
    out.update(rt_boxes=tubes, rt_kps=tkps, rt_props=tprops)
    entry = dict(boxes=tubes.copy(), gt_classes=np.ones((n,), np.int32), is_crowd=np.zeros((n,), np.bool_),
                 gt_overlaps=scipy.sparse.csr_matrix(tov), box_to_gt_ind_map=np.arange(n, dtype=np.int32),
                 gt_keypoints=tkps.copy(), seg_areas=np.zeros((n,), np.float32), segms=[[] for _ in range(n)], height=H, width=W)
    roidb = [entry]
    jd._merge_proposal_boxes_into_roidb(roidb, [tprops])
    jd._add_class_assignments(roidb)
    out['rt_merged_max_overlaps'] = roidb[0]['max_overlaps']
    out['rt_merged_b2g'] = roidb[0]['box_to_gt_ind_map']
    npr.seed(79)
    np.random.seed(79)
    sb = rfr._sample_rois(roidb[0], 1.25, 0)
    for k, v in sb.items():
        out['rt_s_' + k] = np.asarray(v)
    # tube RPN labels: anchors with time_dim T, per-frame visibility of the ground-truth tracks in the inside weights (rpn.py:285-300)
    rrpn._threadlocal_foa.cache = {}        # (the reference memoises fields by stride / sizes / ratios only -- not by time_dim)
    tfoas = [rrpn._get_field_of_anchors(2. ** lvl, (cfg.FPN.RPN_ANCHOR_START_SIZE * 2. ** (lvl - cfg.FPN.RPN_MIN_LEVEL),),
                                        cfg.FPN.RPN_ASPECT_RATIOS, T) for lvl in range(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.RPN_MAX_LEVEL + 1)]
    vis = np.array([[1, 1, 1], [1, 0, 1], [1, 1, 0], [1, 1, 1]], dtype=bool)
    npr.seed(80)
    tblobs = rrpn._get_rpn_blobs(float(H), float(W), tfoas, np.concatenate([f.field_of_anchors for f in tfoas]), tubes, vis)
    out['rt_vis'] = vis
    for i, b in enumerate(tblobs):
        for k, v in b.items():
            if 'vis' not in k:
                out['rt_%s_fpn%d' % (k, i + 2)] = v
    cfg.MODEL.VIDEO_ON, cfg.VIDEO.NUM_FRAMES, cfg.VIDEO.NUM_FRAMES_MID = False, 1, 1
    np.savez_compressed(os.path.join(HERE, 'reference_roi_data.npz'), **out)
    print('wrote', os.path.join(HERE, 'reference_roi_data.npz'), len(out), 'arrays')
    golden_posetrack_json(cfg)


def golden_posetrack_json(cfg):
    """lib/core/mpii_eval_engine.py:_convert_data_to_annorect_struct on seeded detections, for every KP_CONF_TYPE ->
    tests/golden/reference_posetrack_annorect.json (inputs included)."""
    import json
    sys.modules.setdefault('tqdm', types.ModuleType('tqdm'))
    sys.modules['tqdm'].tqdm = lambda x, **k: x
    import core.mpii_eval_engine as ref
    rs = np.random.RandomState(5)
    n = 6
    boxes = np.hstack([rs.uniform(0, 300, (n, 4)), rs.uniform(0.2, 1.0, (n, 1))]).astype(np.float32)
    poses = [np.vstack([rs.uniform(0, 300, (2, 17)), rs.uniform(-2, 8, (1, 17)), rs.uniform(0, 1, (1, 17))]).astype(np.float32)
             for _ in range(n)]
    tracks = [int(v) for v in rs.randint(0, 50, n)]
    out = {'boxes': boxes.tolist(), 'poses': [p.tolist() for p in poses], 'tracks': tracks, 'cases': []}
    for conf_type in ('global', 'local', 'scaled'):
        for thr in (-float('inf'), 1.95):
            cfg.TRACKING.KP_CONF_TYPE = conf_type
            cfg.EVAL.EVAL_MPII_KPT_THRESHOLD = thr
            res = ref._convert_data_to_annorect_struct(boxes, poses, tracks)
            out['cases'].append({'conf_type': conf_type, 'thr': thr if thr > -1e30 else None, 'annorect': res})
    out['empty'] = ref._convert_data_to_annorect_struct(np.zeros((0, 5), np.float32), [], [])
    with open(os.path.join(HERE, 'reference_posetrack_annorect.json'), 'w') as f:
        json.dump(out, f, default=float)   # 'local' / 'scaled' scores are np.float32 in the reference
    print('wrote reference_posetrack_annorect.json', len(out['cases']), 'cases')


def tracker_case(seed, T, frames_per_video, algo):
    """Seeded detections of a few videos in the detections.pkl layout (test_engine.py:199-204), clip-named roidb entries in SHUFFLED order
    inside each video (the tracker sorts them by key-frame path, tracking_engine.py:681), with low-confidence, tiny, out-of-image and
    empty frames in the mix.  Returns (json_data, dets)."""
    rs = np.random.RandomState(seed)
    json_data, boxes_all, keyps_all = [], [], []
    for v, nf in enumerate(frames_per_video):
        n_p = int(rs.randint(2, 7))
        pos = np.stack([rs.uniform(50, 1100, n_p), rs.uniform(50, 600, n_p)], axis=1)
        size = rs.uniform(40, 220, (n_p, 2))
        vel = rs.uniform(-25, 25, (n_p, 2))
        order = rs.permutation(nf)                        # entries of one video arrive out of order
        per_frame = {}
        for f in range(nf):
            p = pos + vel * f
            keep = rs.uniform(size=n_p) > 0.15            # a person is missed now and then
            if f == nf // 2 and nf > 3:
                keep[:] = False                           # one frame without any detection
            rows = []
            for i in np.where(keep)[0]:
                tube = []
                for t in range(T):
                    j = rs.uniform(-3, 3, 4)
                    tube += [p[i, 0] + j[0] + 2 * t, p[i, 1] + j[1], p[i, 0] + size[i, 0] + j[2] + 2 * t, p[i, 1] + size[i, 1] + j[3]]
                score = rs.choice([0.99, 0.95, 0.91, 0.9, 0.89, 0.5])
                rows.append(tube + [score])
            if rs.uniform() < 0.3:                        # a tiny box and one hanging out of the image
                rows.append([10.0, 10.0, 15.0, 16.0] * T + [0.97])
                rows.append([1250.0, 690.0, 1400.0, 800.0] * T + [0.96])
            b = np.array(rows, dtype=np.float32).reshape(-1, 4 * T + 1)
            b = b[rs.permutation(len(b))]
            k = [rs.uniform(0, 1, (4, 17 * T)).astype(np.float32) for _ in range(len(b))]
            per_frame[f] = (b, k)
        for f in order:
            names = ['/data/vid%02d/%05d.jpg' % (v, min(max(f - T // 2 + j, 0), nf - 1)) for j in range(T)]
            json_data.append({'image': names if T > 1 else names[0], 'height': 720, 'width': 1280})
            boxes_all.append(per_frame[f][0])
            keyps_all.append(per_frame[f][1])
    dets = {'all_boxes': [[], boxes_all], 'all_keyps': [[], keyps_all], 'all_segms': [[], [None] * len(boxes_all)], 'cfg': None}
    return json_data, dets


def golden_tracker(cfg):
    """lib/core/tracking_engine.py of the REAL reference (its imports of the drawing module stubbed, py2's list-returning range given
    back to it): _center_detections (:751-755), _prune_bad_detections (:731-748), compute_matches_tracks (:669-708) with the Hungarian
    and the greedy matcher on seeded detections -> tests/golden/reference_tracker.json (generator arguments + expected outputs)."""
    import json
    sys.modules['utils.vis'] = _Stub('utils.vis')       # matplotlib drawing only: not on the tracker's arithmetic path
    import core.tracking_engine as ref
    ref.range = lambda *a: list(range(*a))              # (py2 semantics: `range(...) + [-1]` in _center_boxes)
    ref.tqdm = lambda it, **k: it
    ref._summarize_track_stats = lambda *a, **k: None   # (prints only)
    cases = []
    for seed, T, frames, algo in ((11, 1, (12, 1, 9), 'hungarian'), (12, 1, (7, 15), 'greedy'), (13, 3, (10, 6, 2), 'hungarian'),
                                  (14, 1, (40,), 'hungarian'), (20, 4, (9, 8, 3), 'hungarian')):     # (seed 20: naming a clip by its FIRST frame instead of its centre frame changes the ids)
        cfg.TRACKING.BIPARTITE_MATCHING_ALGO = algo
        cfg.TRACKING.DISTANCE_METRICS = ('bbox-overlap', 'cnn-cosdist', 'pose-pck')
        cfg.TRACKING.DISTANCE_METRIC_WTS = (1.0, 0.0, 0.0)
        cfg.KRCNN.NUM_KEYPOINTS = 17
        json_data, dets = tracker_case(seed, T, frames, algo)
        if cfg.TRACKING.KEEP_CENTER_DETS_ONLY:
            ref._center_detections(dets)
        centred = [b.copy() for b in dets['all_boxes'][1]]
        dets = ref._prune_bad_detections(dets, json_data, cfg.TRACKING.CONF_FILTER_INITIAL_DETS)
        out = ref.compute_matches_tracks(json_data, dets, None)
        cases.append({'seed': seed, 'T': T, 'frames_per_video': list(frames), 'algo': algo,
                      'conf': float(cfg.TRACKING.CONF_FILTER_INITIAL_DETS),
                      'centred_shapes': [list(b.shape) for b in centred],
                      'pruned_boxes': [np.asarray(b, dtype=np.float64).round(4).tolist() for b in out['all_boxes'][1]],
                      'pruned_pose_counts': [len(k) for k in out['all_keyps'][1]],
                      'pose_shapes': [list(k[0].shape) if len(k) else None for k in out['all_keyps'][1]],
                      'tracks': [[int(t) for t in tr] for tr in out['all_tracks'][1]]})
    with open(os.path.join(HERE, 'reference_tracker.json'), 'w') as f:
        json.dump({'cases': cases}, f)
    print('wrote reference_tracker.json', len(cases), 'cases,', sum(len(c['tracks']) for c in cases), 'frames')


BLOB_CASES = (('b2d_fpn', dict(video=False, fpn=True, T=1, shapes=((37, 50), (40, 45)))),
              ('b3d_fpn_T4', dict(video=True, fpn=True, T=4, shapes=((20, 40),) * 8)),
              ('b3d_c4_T3', dict(video=True, fpn=False, T=3, shapes=((33, 41),) * 3)),
              ('b2d_c4', dict(video=False, fpn=False, T=1, shapes=((20, 31), (25, 30), (25, 31)))))
PREP_CASES = ((600, 800, 800, 1333), (720, 1280, 800, 1333), (1080, 1920, 800, 1333), (96, 128, 96, 1000), (100, 1000, 800, 1333),
              (480, 854, 600, 1000), (500, 375, 800, 1333), (333, 500, 500, 833))


def blob_case_images(name, shapes):
    rs = np.random.RandomState(sum(map(ord, name)))
    return [rs.randint(-120, 130, sh + (3,)).astype(np.float32) for sh in shapes]       # (integer-valued: the fixture compresses)


def golden_blob(cfg):
    """lib/utils/blob.py of the REAL reference: im_list_to_blob (:40-68: common size, FPN.COARSEST_STRIDE padding, NCHW, time axis of video
    models through utils/image.move_batch_to_time) and the scale rule + argument order of prep_im_for_blob (:71-90; cv2 is absent here, so
    its resize is replaced by a recorder: the golden holds the (fx, fy, interpolation) it was called with and the mean-subtracted image
    it was handed) -> tests/golden/reference_blob.npz."""
    import utils.blob as ref
    out = {}
    for name, c in BLOB_CASES:
        cfg.MODEL.VIDEO_ON, cfg.FPN.FPN_ON, cfg.VIDEO.NUM_FRAMES = c['video'], c['fpn'], c['T']
        out[name] = ref.im_list_to_blob(blob_case_images(name, c['shapes']))
    calls = []

    def fake_resize(im, dsize, dst, fx=None, fy=None, interpolation=None):
        calls.append((fx, fy, interpolation, im.copy()))
        return im
    ref.cv2.resize, ref.cv2.INTER_LINEAR = fake_resize, 1
    rs = np.random.RandomState(5)
    means = np.array([[[102.9801, 115.9465, 122.7717]]])
    for k, (h, w, target, max_size) in enumerate(PREP_CASES):
        im = rs.randint(0, 255, (8, 8, 3)).astype(np.uint8)
        im = np.ascontiguousarray(np.broadcast_to(im[:1, :1], (h, w, 3)))         # (only the shape matters to the rule; tiny content)
        del calls[:]
        _, scales = ref.prep_im_for_blob(im, means, [target], max_size)
        fx, fy, interp, handed = calls[0]
        out['prep%d' % k] = np.array([h, w, target, max_size, scales[0], fx, fy, interp], dtype=np.float64)
        out['prep%d_pixel' % k] = np.concatenate([im[0, 0].astype(np.float64), handed[0, 0].astype(np.float64)])
    np.savez_compressed(os.path.join(HERE, 'reference_blob.npz'), **out)
    print('wrote reference_blob.npz', len(out), 'arrays')


def decode_case_inputs(seed, n, K, M=56):
    """Seeded heatmaps + rois for the decode golden: rois narrower than a pixel (the max(., 1) clamp), fractional sizes (ceil), a map
    whose maximum is attained twice (the first position wins) and one that is constant."""
    rs = np.random.RandomState(seed)
    maps = (rs.randn(n, K, M, M) * 2).astype(np.float32)
    maps[0, 1] = np.round(maps[0, 1])                     # plateaus after resampling are rare; ties in the source are not
    maps[1, 2, :, :] = 0.25                               # a constant map: argmax = first cell
    xy = rs.uniform(0, 200, (n, 2)).astype(np.float32)
    wh = rs.uniform(0.3, 90, (n, 2)).astype(np.float32)
    wh[0] = (0.4, 37.5)
    wh[1] = (56.0, 0.9)
    wh[2] = (12.0, 12.0)
    return maps, np.hstack((xy, xy + wh)).astype(np.float32)


def tube_decode_inputs(seed, n, K, T):
    rs = np.random.RandomState(seed)
    maps = (rs.randn(n, K * T, 56, 56) * 2).astype(np.float32)
    xy = rs.uniform(0, 150, (n, 2)).astype(np.float32)
    rois = np.zeros((n, 4 * T), np.float32)
    for t in range(T):
        j = rs.uniform(-5, 5, (n, 2)).astype(np.float32)
        rois[:, 4 * t:4 * t + 2] = xy + j
        rois[:, 4 * t + 2:4 * t + 4] = xy + j + rs.uniform(3, 70, (n, 2)).astype(np.float32)
    return maps, rois


def golden_decode(cfg):
    """lib/utils/keypoints.py:94-149 heatmaps_to_keypoints and :210-216 scores_to_probs of the REAL reference, with cv2.resize -- OpenCV
    is not in this image -- replaced by the oracle's restatement of INTER_CUBIC (oracle/resize.py, pinned by exact-rational known answers):
    everything AROUND the resampler is the reference's own code: roi size / ceil / min-size rules, the argmax convention, the
    (x + 0.5) * correction + offset formula, logit and spatial-softmax probability -> tests/golden/reference_decode.npz."""
    from oracle import resize as oresize
    import utils.keypoints as ref
    ref.cv2.INTER_CUBIC = 2

    def fake_resize(im, dsize, interpolation=None):
        assert interpolation == 2
        return oresize.resize_cubic(im, dsize=(int(dsize[0]), int(dsize[1])))
    ref.cv2.resize = fake_resize
    out = {}
    for name, (seed, n, K, min_size) in {'dec_k17': (31, 5, 17, 0), 'dec_k17_min40': (31, 5, 17, 40), 'dec_k3_min8': (32, 4, 3, 8)}.items():
        cfg.KRCNN.NUM_KEYPOINTS, cfg.KRCNN.INFERENCE_MIN_SIZE = K, min_size
        maps, rois = decode_case_inputs(seed, n, K)
        out[name] = ref.heatmaps_to_keypoints(maps, rois)
        out[name + '_cfg'] = np.array([seed, n, K, min_size])
    sc = (np.random.RandomState(33).randn(4, 9, 7) * 3).astype(np.float32)
    out['probs_in'], out['probs_out'] = sc, ref.scores_to_probs(sc.copy())
    # core/test.py:865-894 keypoint_results on TUBE detections (T = 2: one decode per frame with that frame's box, rows concatenated along
    # the keypoint axis) and :77-121 _get_rois_blob (the keypoint net's roi blob: level 0, boxes x the image scale)
    import core.test as ref_test
    T, K = 2, 3
    cfg.KRCNN.NUM_KEYPOINTS, cfg.KRCNN.INFERENCE_MIN_SIZE, cfg.MODEL.NUM_CLASSES, cfg.KRCNN.NMS_OKS = K, 0, 2, False
    maps, rois = tube_decode_inputs(34, 4, K, T)
    kps = ref_test.keypoint_results([[], rois], maps, rois)
    assert len(kps) == 2 and kps[0] == []
    out['tube_keyps'] = np.stack(kps[1])
    out['tube_cfg'] = np.array([34, 4, K, T])
    out['tube_rois_blob'] = ref_test._get_rois_blob(rois, np.array([1.0414]))
    np.savez_compressed(os.path.join(HERE, 'reference_decode.npz'), **out)
    print('wrote reference_decode.npz', len(out), 'arrays')


def golden_cfg_defaults(cfg):
    """Every key and default of the REAL reference's lib/core/config.py (flattened, JSON) -> tests/golden/reference_cfg_defaults.json.
    Must run first: the other generators edit cfg."""
    import json

    def flat(d, pre=''):
        out = {}
        for k, v in d.items():
            if isinstance(v, dict):
                out.update(flat(v, pre + str(k) + '.'))
            else:
                if isinstance(v, bytes):
                    v = v.decode()
                if isinstance(v, np.ndarray):
                    v = v.tolist()
                out[pre + str(k)] = list(v) if isinstance(v, tuple) else v
        return out
    with open(os.path.join(HERE, 'reference_cfg_defaults.json'), 'w') as f:
        json.dump(flat(cfg), f, indent=0, sort_keys=True)
    print('wrote reference_cfg_defaults.json', len(flat(cfg)), 'keys')


def golden_cfg_files(cfg):
    """The EFFECTIVE configuration of every shipped yaml (configs/video/*/*.yaml) as the REAL reference computes it: cfg_from_file
    (yaml 1.1 load, type rules of _merge_a_into_b, the TIME_KERNEL_DIM / int-to-dict rules) + assert_and_infer_cfg -> the keys that
    differ from the defaults, per file -> tests/golden/reference_cfg_files.json.  Must run before any generator that edits cfg."""
    import copy
    import glob
    import json
    import core.config as rc

    def flat(d, pre=''):
        out = {}
        for k, v in d.items():
            if isinstance(v, dict):
                out.update(flat(v, pre + str(k) + '.'))
            else:
                if isinstance(v, bytes):
                    v = v.decode()
                if isinstance(v, np.ndarray):
                    v = v.tolist()
                out[pre + str(k)] = list(v) if isinstance(v, tuple) else v
        return out
    import yaml as _yaml
    _load = _yaml.load
    _yaml.load = lambda f, Loader=None: _load(f, Loader=Loader or _yaml.Loader)   # (the PyYAML of the reference's day: no Loader argument = the full Loader)
    default = copy.deepcopy(rc.cfg)
    base = flat(default)
    res = {}
    for f in sorted(glob.glob('/root/reference/configs/video/*/*.yaml')):
        for k in list(rc.cfg.keys()):
            rc.cfg[k] = copy.deepcopy(default[k])
        rc.cfg_from_file(f)
        rc.assert_and_infer_cfg()
        now = flat(rc.cfg)
        res[os.path.relpath(f, '/root/reference/configs')] = {k: v for k, v in now.items() if k not in base or base[k] != v}
    for k in list(rc.cfg.keys()):
        rc.cfg[k] = copy.deepcopy(default[k])
    _yaml.load = _load
    with open(os.path.join(HERE, 'reference_cfg_files.json'), 'w') as f:
        json.dump(res, f, indent=0, sort_keys=True)
    print('wrote reference_cfg_files.json', len(res), 'files,', sum(len(v) for v in res.values()), 'non-default keys')


BUILDER_CASES = (('fpn3d_r18_T8', 'fpn3d_kps_cfg', dict(arch='18', T=8)), ('fpn3d_r50_T8', 'fpn3d_kps_cfg', dict(arch='50', T=8)),
                 ('fpn3d_r101_T4_avg', 'fpn3d_kps_cfg', dict(arch='101', T=4, link='avg')), ('fpn2d_r50', 'fpn2d_kps_cfg', dict(arch='50')))
# (the non-FPN tube model's RPN outputs, model_builder.py:540-600, go through a chain of Caffe2 shape ops -- GetShapeDimIdx, Reshape,
#  ExpandDims -- that this package folds into the proposal kernel's addressing: not comparable op by op, covered numerically instead)


def net_signature(net):
    """[type, inputs, outputs, args] of every recorded op, JSON-able"""
    def plain(v):
        if hasattr(v, 'tolist'):
            return v.tolist()
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        if isinstance(v, dict):
            return {str(k): plain(x) for k, x in sorted(v.items())}
        return v if isinstance(v, (int, float, bool, str, type(None))) else str(v)
    return [[o.type, [str(b) for b in o.inputs], [str(b) for b in o.outputs], plain(dict(o.args))] for o in net.ops]


def golden_builders(cfg):
    """The REFERENCE's own graph builders -- lib/modeling/model_builder.py create() -> keypoint_rcnn -> build_generic_fast_rcnn_model
    (:179-306) with ResNet3D.py / ResNet.py / FPN3D.py / FPN.py / head_builder.py / keypoint_rcnn_heads.py and the output functions
    (:426-478, :500-609, :755-870) -- executed on a RECORDER: this package's DetectionModelHelper (the mirror of the helper API those
    builders call) plus the few Caffe2-only calls they make around it.  What is pinned is the wiring: every conv / affine / pool / sum /
    roi transform / output op, its inputs, outputs, kernel, stride, pad, init spec, in the reference's order, and every parameter name
    -> tests/golden/reference_builder_nets.json.gz.  Replaced, not executed: build_data_parallel_model (one replica: no name scopes, no
    gradient ops), the NetDef surgery of get_suffix_net (same split, on the recorded list), the loss functions (their fused
    counterparts are checked against autograd on the GPU)."""
    import json
    import queue
    sys.modules['Queue'] = queue
    sys.modules['caffe2.python.cnn'].CNNModelHelper = type('CNNModelHelper', (object,), {})
    sys.modules['caffe2.python.core'].BlobReference = type('BlobReference', (object,), {})     # (blobs are plain names on the recorder)
    import modeling.model_builder as rmb
    from detectandtrack_amd.core.config import cfg as my_cfg, reset_cfg, cfg_from_cfg, assert_and_infer_cfg
    from detectandtrack_amd.modeling.detector import DetectionModelHelper, Net, Op
    from tests import model_util

    class RecNet(Net):
        @property
        def op(self):
            return self.ops

        def GetBlobRef(self, name):
            return name

        def __getattr__(self, name):        # any other raw Caffe2 op the reference emits through model.net.<Op>: recorded as is
            if name.startswith('_') or not name[0].isupper():
                raise AttributeError(name)

            def rec(blobs_in, blobs_out=None, **kw):
                ins = [blobs_in] if isinstance(blobs_in, str) else list(blobs_in)
                outs = [] if blobs_out is None else ([blobs_out] if isinstance(blobs_out, str) else list(blobs_out))
                self.ops.append(Op(name, ins, outs, **kw))
                return outs[0] if len(outs) == 1 else outs
            return rec

    class Recorder(DetectionModelHelper):
        def __init__(self, **kw):
            DetectionModelHelper.__init__(self, **kw)
            net = RecNet(self.net.name, self)
            net.ops = self.net.ops
            self.net = net

        def ConstantFill(self, blobs_in, blob_out, **kw):       # (:192-193 'zero' / 'minus1': constants of the Caffe2 runtime)
            return blob_out

        def __getattr__(self, name):        # helper-level Caffe2 ops this package has no mirror of (the loss functions' Accuracy, ...)
            if name.startswith('_') or not name[0].isupper():
                raise AttributeError(name)
            return getattr(self.net, name)

        # the time -> channel move and its inverse around the reference's 2D UpsampleNearest (FPN3D.py:205-219) cancel: the blob keeps
        # the name the inverse gives it
        def GetTemporalDim(self, blob):
            return ('T', str(blob))

        def MoveTimeToChannelDim(self, blob_in, blob_out=None):
            if blob_out is not None and str(blob_out).endswith('_time2ch'):
                return blob_in
            return DetectionModelHelper.MoveTimeToChannelDim(self, blob_in, blob_out)

        def MoveTimeToChannelDimInverse(self, blob_in, blob_out, temporal_dim):
            op = self.net.producer(blob_in)
            op.outputs = [blob_out if str(o) == str(blob_in) else o for o in op.outputs]
            return blob_out

    def split(name, prefix_ops, net, outputs):                  # get_suffix_net (:994-1021) on the recorded list
        assert [(o.type, o.outputs) for o in prefix_ops] == [(o.type, o.outputs) for o in net.ops[:len(prefix_ops)]]     # (deep copies)
        new = Net(name, net._helper)
        new.ops = net.ops[len(prefix_ops):]
        return new, outputs
    rmb.get_suffix_net = split
    rmb.build_data_parallel_model = lambda model, build: build(model)
    rmb.init_model = lambda name, train, init_params=None: Recorder(name=name, train=train, num_classes=cfg.MODEL.NUM_CLASSES,
                                                                    init_params=init_params or train)

    def mirror(dst, src):
        for k, v in src.items():
            if k == 'HIP':
                continue
            if isinstance(v, dict):
                if k in dst:
                    mirror(dst[k], v)
            elif k in dst:
                dst[k] = v
    out = {}
    for name, fn, kw in BUILDER_CASES:
        for train in (False, True):
            c = getattr(model_util, fn)(**kw)
            reset_cfg()
            cfg_from_cfg(c)
            if train:
                my_cfg.TRAIN.DATASET = 'synthetic'
            assert_and_infer_cfg()
            mirror(cfg, my_cfg)
            m = rmb.create(cfg.MODEL.TYPE, train=train)
            rec = {'cfg_fn': fn, 'cfg_kw': kw, 'train': train, 'params': [str(p) for p in m.params]}
            if train:
                rec['net'] = net_signature(m.net)
            else:
                bbox = m.net._net                                # (what the reference restores as the primary net, :266-267)
                main = Net('net', m)
                main.ops = m.net.ops[:len(bbox.ops)]
                rec['net'], rec['keypoint_net'] = net_signature(main), net_signature(m.keypoint_net)
                rec['conv_body_net_ops'] = len(m.conv_body_net.ops)
                # what ONE recorded RoIFeatureTransform stands for: the reference helper's own expansion (detector.py:216-310 run unbound
                # on a recorder) with the arguments the head builders passed -- per level RoIAlign(level blob, rois_fpnK), Concat,
                # BatchPermutation with the restore indices
                import types
                import modeling.detector as rdet
                rec['roi_transforms'] = []
                for o in main.ops + m.keypoint_net.ops:
                    if o.type != 'RoIFeatureTransform':
                        continue
                    n_feat = o.args['n_feat']
                    r = Recorder(name='r', train=False, num_classes=cfg.MODEL.NUM_CLASSES)
                    r._do_roi_transform = types.MethodType(rdet.DetectionModelHelper._do_roi_transform, r)
                    ret = rdet.DetectionModelHelper.RoIFeatureTransform(
                        r, [str(b) for b in o.inputs[:n_feat]][::-1], str(o.outputs[0]), blob_rois=str(o.inputs[n_feat]), method='RoIAlign',
                        resolution=o.args['resolution'], spatial_scale=list(o.args['scales'])[::-1], sampling_ratio=o.args['sampling_ratio'])
                    rec['roi_transforms'].append({'fused': net_signature(types.SimpleNamespace(ops=[o]))[0], 'expansion': net_signature(r.net),
                                                  'returns': str(ret)})
            out[name + ('_train' if train else '')] = rec
    # the keypoint OUTPUT function on a 3D head, both settings of KRCNN.NO_3D_DECONV_TIME_TO_CH (model_builder.py:755-870 run on the
    # recorder with the arguments build_generic_fast_rcnn_model passes for a T = 3 tube head): False is the reference DEFAULT
    # (core/config.py:472) -- time -> channels, ConvTranspose with group = time_dim over dim*T -> K*T channels, the bilinear deconv on
    # K*T maps; True is what the shipped 3D configs set (time -> batch ... batch -> time, time -> channels)
    heat = {}
    for no_t2c in (False, True):
        reset_cfg()
        cfg_from_cfg(model_util.c4_tube_kps_cfg(T=3, deconv='time_to_batch' if no_t2c else 'grouped'))
        assert_and_infer_cfg()
        mirror(cfg, my_cfg)
        assert cfg.KRCNN.NO_3D_DECONV_TIME_TO_CH == no_t2c
        r = Recorder(name='heat', train=False, num_classes=cfg.MODEL.NUM_CLASSES)
        ret = rmb.add_heatmap_outputs(r, 'conv_fcn8', cfg.KRCNN.CONV_HEAD_DIM, 3, True)
        heat['no_3d_deconv_time_to_ch_%s' % no_t2c] = {'ops': net_signature(r.net), 'returns': str(ret), 'params': [str(p) for p in r.params]}
    with open(os.path.join(HERE, 'reference_heatmap_outputs.json'), 'w') as f:
        json.dump(heat, f, sort_keys=True, indent=1)
    print('wrote reference_heatmap_outputs.json', {k: [o[0] for o in v['ops']] for k, v in heat.items()})
    reset_cfg()
    import gzip
    with gzip.GzipFile(os.path.join(HERE, 'reference_builder_nets.json.gz'), 'wb', mtime=0) as f:
        f.write(json.dumps(out, sort_keys=True).encode())
    print('wrote reference_builder_nets.json.gz', {k: len(v['net']) for k, v in out.items()})


def weights_case(T=3):
    """A crafted checkpoint + the model it is loaded into (name -> shape, in model order; name -> initialised value): an exact match, a 2D
    kernel to inflate into a 3D one, a 2D tensor repeated T times along its output axis, a shape that cannot be inflated, a parameter the
    file lacks, the `_[xyz]_foo <- foo` sharing rule with and without the full name in the file, momentum blobs, BN statistics and a
    blob no parameter uses."""
    rs = np.random.RandomState(41)
    f32 = lambda *sh: rs.randn(*sh).astype(np.float32)
    shapes = [('conv1_w', (4, 3, 1, 3, 3)), ('res2_w', (4, 4, 3, 3, 3)), ('res2_b', (4,)), ('pred_w', (6 * T, 8)), ('pred_b', (6 * T,)),
              ('odd_w', (5, 7)), ('new_head_w', (3, 3)), ('_[pose]_fc_w', (2, 5)), ('_[mask]_fc_w', (2, 5)), ('fc_w', (2, 5))]
    init = {n: f32(*sh) for n, sh in shapes}
    blobs = {'conv1_w': f32(4, 3, 1, 3, 3), 'res2_w': f32(4, 4, 3, 3), 'res2_b': f32(4), 'pred_w': f32(6, 8), 'pred_b': f32(6),
             'odd_w': f32(4, 7), 'fc_w': f32(2, 5), '_[mask]_fc_w': f32(2, 5), 'conv1_w_momentum': f32(4, 3, 1, 3, 3),
             'res2_b_momentum': f32(4), 'res2_bn_rm': f32(4), 'res2_bn_riv': f32(4), 'unused_w': f32(2, 2)}
    return shapes, init, blobs


def golden_weights(cfg):
    """lib/utils/net.py:163-249 initialize_gpu_0_from_weights_file ITSELF (with :72-161 inflate_weights) against a dict-backed stand-in
    of the Caffe2 workspace: which blob of a checkpoint lands in which parameter, inflated how, which momentum blobs are restored, what
    is preserved -> tests/golden/reference_weights_load.npz (every FeedBlob the reference makes, for both kinds of file name)."""
    import pickle as _pickle
    import tempfile
    import utils.net as rnet
    shapes, init, blobs = weights_case()
    store = {}

    class WS(object):
        @staticmethod
        def Blobs():
            return list(store.keys())

        @staticmethod
        def FetchBlob(name):
            return store[str(name)]

        @staticmethod
        def FeedBlob(name, arr):
            store[str(name)] = np.asarray(arr)
            fed.append(str(name))

    class Ctx(object):
        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    class Core(object):
        NameScope = DeviceScope = staticmethod(lambda *a, **k: Ctx())
        DeviceOption = staticmethod(lambda *a, **k: None)
        ScopedName = staticmethod(lambda n: 'gpu_0/' + str(n))

    class Pickle(object):           # (py2 opened pickles in text mode, :167)
        load = staticmethod(lambda f: _pickle.load(open(f.name, 'rb')))
    rnet.workspace, rnet.core, rnet.pickle = WS, Core, Pickle
    rnet.utils.blob.unscope_name = lambda n: n[n.rfind('/') + 1:]
    out = {}
    cfg.VIDEO.WEIGHTS_INFLATE_MODE = 'center-only'
    d = tempfile.mkdtemp()
    for tag, fname in (('resume', 'model_iter99.pkl'), ('first', 'R-50_trainedCOCO.pkl')):
        path = os.path.join(d, fname)
        with open(path, 'wb') as f:
            _pickle.dump({'blobs': blobs}, f, protocol=2)
        store.clear()
        store.update({'gpu_0/' + n: v.copy() for n, v in init.items()})
        fed = []
        model = type('M', (object,), {'params': ['gpu_0/' + n for n, _ in shapes]})()
        rnet.initialize_gpu_0_from_weights_file(model, path)
        out[tag + '_fed'] = np.array(fed)
        for n in fed:
            out[tag + ':' + n] = store[n]
    np.savez_compressed(os.path.join(HERE, 'reference_weights_load.npz'), **out)
    print('wrote reference_weights_load.npz', len(out), 'arrays;', list(out['resume_fed']))


CLIP_CASES = ((3, 1, (7, 3)), (4, 1, (7, 3)), (8, 1, (10, 2, 1)), (5, 2, (9, 4)), (4, 3, (11,)))


def golden_clips(cfg):
    """lib/utils/video.py:149-201 get_clip ITSELF on synthetic per-frame roidbs (its tube-building _combine_clips replaced by the one
    line of it that matters to inference: the frame list, :70): which frames make the clip of every key frame, incl. the border
    replication and VIDEO.TIME_INTERVAL -> tests/golden/reference_clips.json."""
    import json
    import types
    import utils.video as rv
    rv.tqdm = lambda it, **k: it
    rv._combine_clips = lambda entry: {'image': [c['image'] for c in entry['clip_ids']]}
    ds = type('DS', (object,), {'frames_from_video': False})()
    out = []
    for T, step, videos in CLIP_CASES:
        cfg.VIDEO.NUM_FRAMES, cfg.VIDEO.NUM_FRAMES_MID, cfg.VIDEO.TIME_INTERVAL = T, T, step
        roidb = [{'image': '/data/vid%02d/%06d.jpg' % (v, f + 1), 'dataset': ds, 'flipped': False} for v, n in enumerate(videos) for f in range(n)]
        clips = rv.get_clip(roidb)
        out.append({'T': T, 'time_interval': step, 'videos': list(videos),
                    'clips': [[int(os.path.splitext(os.path.basename(p))[0]) for p in c['image']] for c in clips]})
    cfg.VIDEO.NUM_FRAMES, cfg.VIDEO.NUM_FRAMES_MID, cfg.VIDEO.TIME_INTERVAL = 1, 1, 1
    with open(os.path.join(HERE, 'reference_clips.json'), 'w') as f:
        json.dump(out, f)
    print('wrote reference_clips.json', [(c['T'], c['time_interval'], len(c['clips'])) for c in out])


KPT_NAMES = ['nose', 'head_bottom', 'head_top', 'left_ear', 'right_ear', 'left_shoulder', 'right_shoulder', 'left_elbow', 'right_elbow',
             'left_wrist', 'right_wrist', 'left_hip', 'right_hip', 'left_knee', 'right_knee', 'left_ankle', 'right_ankle']


def synthetic_posetrack_json(seed=23):
    """A small COCO-format annotation file shaped like the PoseTrack lists the reference reads (lib/datasets/lists/PoseTrack/v1.0/*.json):
    three videos (5, 4 -- frame 3 missing -- and 3 frames), persons with track ids, 17 keypoints, head boxes; plus the records the
    reference's sanitiser drops or treats specially: `ignore`, zero area, a one-pixel box, a box that leaves the image, an area below
    TRAIN.GT_MIN_AREA is not used (default -1), a crowd region (RLE dict), a polygon with fewer than three points, an image without
    annotations, image ids out of order in the file."""
    rs = np.random.RandomState(seed)
    images, anns = [], []
    sizes = {'bonn_000001': (480, 640), 'mpii_000002': (360, 540), 'bonn_000003': (480, 854)}
    frames = {'bonn_000001': [1, 2, 3, 4, 5], 'mpii_000002': [1, 2, 4, 5], 'bonn_000003': [7, 8, 9]}
    img_id = 1000
    ann_id = 1
    for vi, (vid, fr) in enumerate(sorted(frames.items())):
        h, w = sizes[vid]
        persons = {tid: (rs.uniform(0.1, 0.6) * w, rs.uniform(0.1, 0.4) * h, rs.uniform(0.1, 0.3) * w, rs.uniform(0.3, 0.55) * h)
                   for tid in range(3 + vi)}
        for f in fr:
            img_id += 7 if f % 2 else -3             # (ids neither contiguous nor monotonic in file order)
            iid = img_id + 100 * vi
            images.append({'id': iid, 'file_name': 'images/%s/%06d.jpg' % (vid, f), 'width': w, 'height': h, 'nframes': len(fr), 'frame_id': f,
                           'is_labeled': bool(f % 2), 'original_file_name': 'images/%s/%08d.jpg' % (vid, f), 'license': 1})
            if vid == 'bonn_000003' and f == 8:
                continue                              # an image without annotations
            for tid, (x, y, bw, bh) in persons.items():
                if (tid + f) % 4 == 0:
                    continue                          # the person is not in this frame
                dx, dy = rs.uniform(-6, 6, 2)
                bx = [float(np.round(x + dx * f, 2)), float(np.round(y + dy, 2)), float(np.round(bw, 2)), float(np.round(bh, 2))]
                kp = []
                for k in range(17):
                    v = int(rs.randint(0, 3))
                    kp += [int(bx[0] + rs.uniform(0, 1) * bx[2]) if v else 0, int(bx[1] + rs.uniform(0, 1) * bx[3]) if v else 0, v]
                anns.append({'id': ann_id, 'image_id': iid, 'category_id': 1, 'bbox': bx, 'area': float(np.round(bx[2] * bx[3], 2)), 'iscrowd': 0,
                             'keypoints': kp, 'num_keypoints': int(sum(1 for q in kp[2::3] if q > 0)), 'track_id': tid,
                             'head_box': [bx[0] + 2, bx[1] + 1, bx[0] + bx[2] / 3, bx[1] + bx[3] / 5],
                             'segmentation': [[bx[0], bx[1], bx[0] + bx[2], bx[1], bx[0] + bx[2], bx[1] + bx[3]], [1.0, 2.0, 3.0, 4.0]]})
                ann_id += 1
        # the special records, on the video's first image
        first = [im for im in images if ('/%s/' % vid) in im['file_name']][0]['id']
        zero_kp = [0] * 51
        specials = [dict(bbox=[10., 10., 50., 80.], area=4000., ignore=1), dict(bbox=[20., 20., 40., 40.], area=0.),
                    dict(bbox=[30., 30., 1., 60.], area=60.), dict(bbox=[w - 30., h - 40., 90., 120.], area=10800.),
                    dict(bbox=[5., 5., 200., 150.], area=30000., iscrowd=1, segmentation={'counts': 'abc', 'size': [h, w]}),
                    dict(bbox=[-15., -8., 60., 70.], area=4200.)]
        for k, sp in enumerate(specials):
            a = {'id': ann_id, 'image_id': first, 'category_id': 1, 'iscrowd': 0, 'keypoints': list(zero_kp), 'num_keypoints': 0, 'track_id': 50 + k,
                 'segmentation': []}
            a.update(sp)
            anns.append(a)
            ann_id += 1
    cats = [{'id': 1, 'name': 'person', 'supercategory': 'person', 'keypoints': KPT_NAMES, 'skeleton': [[1, 2], [2, 3]]}]
    return {'images': images, 'annotations': anns, 'categories': cats}


class _StubCOCO(object):
    """The slice of pycocotools.coco.COCO that lib/datasets/json_dataset.py calls (pycocotools is absent): a plain index over the JSON,
    written here independently of the package's CocoIndex."""

    def __init__(self, annotation_file):
        import json
        with open(annotation_file) as f:
            d = json.load(f)
        self._imgs = [(im['id'], im) for im in d['images']]
        self._cats = [(c['id'], c) for c in d['categories']]
        self._anns = d['annotations']

    def getImgIds(self):
        return [i for i, _ in self._imgs]

    def loadImgs(self, ids):
        m = dict(self._imgs)
        return [m[i] for i in ids]

    def getCatIds(self):
        return [i for i, _ in self._cats]

    def loadCats(self, ids):
        m = dict(self._cats)
        return [m[i] for i in ids]

    def getAnnIds(self, imgIds=(), iscrowd=None):
        want = set(imgIds if isinstance(imgIds, (list, tuple)) else [imgIds])
        return [a['id'] for a in self._anns if a['image_id'] in want and (iscrowd is None or a['iscrowd'] == iscrowd)]

    def loadAnns(self, ids):
        m = {a['id']: a for a in self._anns}
        return [m[i] for i in ids]


ROIDB_ARRAYS = ('boxes', 'tracks', 'head_boxes', 'gt_classes', 'seg_areas', 'is_crowd', 'box_to_gt_ind_map', 'gt_keypoints', 'max_classes',
                'max_overlaps', 'track_visible')
ROIDB_SCALARS = ('id', 'width', 'height', 'nframes', 'frame_id', 'is_labeled', 'flipped', 'has_visible_keypoints')


def roidb_record(roidb, out, prefix):
    """Flatten a roidb into named arrays (npz) + a JSON-able list of the non-array fields."""
    meta = []
    for i, e in enumerate(roidb):
        for k in ROIDB_ARRAYS:
            if k in e:
                out['%s/%d/%s' % (prefix, i, k)] = np.asarray(e[k])
        out['%s/%d/gt_overlaps' % (prefix, i)] = e['gt_overlaps'].toarray()
        m = {k: (bool(e[k]) if isinstance(e[k], (bool, np.bool_)) else int(e[k])) for k in ROIDB_SCALARS if k in e and not isinstance(e[k], np.ndarray)}
        m['image'] = e['image']
        m['keys'] = sorted(str(k) for k in e.keys())
        m['n_segms'] = [len(s) for s in e['segms']]
        for k in ('all_frame_ids', 'original_file_name'):
            if k in e:
                m[k] = e[k]
        meta.append(m)
    return meta


DATASET_CLIP_CASES = ((3, 3, 1, False), (3, 1, 1, False), (2, 2, 1, False), (3, 3, 2, False), (3, 3, 1, True), (1, 1, 1, False))


def golden_dataset(cfg):
    """lib/datasets/json_dataset.py ITSELF (JsonDataset.__init__, get_roidb with and without ground truth, proposals from a file) and
    lib/utils/video.py:38-201 (get_video_info, get_clip, _combine_clips -- the tube ground truth) run on tests/golden/synthetic_posetrack.json
    through a stub of the four pycocotools.COCO calls it makes -> tests/golden/reference_json_dataset.npz / .json.  Not pinned: the crowd
    filter (:479-497 calls pycocotools.mask.iou, C code that is not in the tree)."""
    import json
    import pickle as pkl
    sys.modules['pycocotools.coco'].COCO = _StubCOCO
    sys.modules['tqdm'] = types.ModuleType('tqdm')
    sys.modules['tqdm'].tqdm = lambda it, **k: it
    import datasets.json_dataset as rj
    import utils.video as rv
    rv.tqdm = lambda it, **k: it
    ann = os.path.join(HERE, 'synthetic_posetrack.json')
    with open(ann, 'w') as f:
        json.dump(synthetic_posetrack_json(), f, sort_keys=True)
    rj.DATASETS['synthetic_posetrack'] = {rj.IM_DIR: '/data/PoseTrack/', rj.ANN_FN: ann, rj.ANN_DN: '/data/annots'}
    ds = rj.JsonDataset('synthetic_posetrack')
    out, meta = {}, {}
    meta['dataset'] = {'classes': ds.classes, 'num_classes': ds.num_classes, 'keypoints': ds.keypoints, 'num_keypoints': ds.num_keypoints,
                       'keypoint_flip_map': ds.keypoint_flip_map, 'category_to_id_map': ds.category_to_id_map,
                       'person_cat_info_keys': sorted(ds.person_cat_info), 'image_directory': ds.image_directory,
                       'annotation_directory': ds.annotation_directory, 'frames_from_video': bool(ds.frames_from_video)}
    meta['gt'] = roidb_record(ds.get_roidb(gt=True), out, 'gt')
    meta['nogt'] = roidb_record(ds.get_roidb(gt=False), out, 'nogt')
    # proposals from a file (:303-331, :424-476): duplicates, tiny boxes, boxes outside the image, file order != id order
    rs = np.random.RandomState(5)
    base = ds.get_roidb(gt=True)
    props = {'boxes': [], 'ids': [], 'scores': []}
    for e in base[::-1]:
        n = 12
        xy = np.stack([rs.uniform(-20, e['width'] - 40, n), rs.uniform(-20, e['height'] - 40, n)], axis=1)
        b = np.hstack([xy, xy + rs.uniform(0.5, 220, (n, 2))]).astype(np.float32)
        b[3] = b[2]
        if len(e['boxes']):
            b[5] = e['boxes'][0] + 1.5
        props['boxes'].append(b)
        props['ids'].append(e['id'])
        props['scores'].append(rs.uniform(0, 1, n).astype(np.float32))
    pfile = os.path.join(HERE, 'synthetic_posetrack_proposals.pkl')
    with open(pfile, 'wb') as f:
        pkl.dump(props, f, protocol=2)
    rj.pickle = types.SimpleNamespace(load=lambda f: pkl.load(open(pfile, 'rb'), encoding='latin1'))     # (py2 text-mode open(..., 'r'))
    meta['props'] = roidb_record(ds.get_roidb(gt=True, proposal_file=pfile, min_proposal_size=2, proposal_limit=8), out, 'props')
    # clips with tube ground truth
    meta['clips'] = []
    for ci, (T, mid, step, imperfect) in enumerate(DATASET_CLIP_CASES):
        cfg.VIDEO.NUM_FRAMES, cfg.VIDEO.NUM_FRAMES_MID, cfg.VIDEO.TIME_INTERVAL = T, mid, step
        clips = rv.get_clip(ds.get_roidb(gt=True), remove_imperfect=imperfect)
        meta['clips'].append({'T': T, 'mid': mid, 'time_interval': step, 'remove_imperfect': imperfect,
                              'entries': roidb_record(clips, out, 'clips%d' % ci)})
    cfg.VIDEO.NUM_FRAMES, cfg.VIDEO.NUM_FRAMES_MID, cfg.VIDEO.TIME_INTERVAL = 1, 1, 1
    np.savez_compressed(os.path.join(HERE, 'reference_json_dataset.npz'), **out)
    with open(os.path.join(HERE, 'reference_json_dataset.json'), 'w') as f:
        json.dump(meta, f, sort_keys=True)
    print('wrote synthetic_posetrack.json, synthetic_posetrack_proposals.pkl, reference_json_dataset.npz (%d arrays) / .json; roidb %d entries, clips %s'
          % (len(out), len(meta['gt']), [len(c['entries']) for c in meta['clips']]))


def golden_postproc(cfg):
    """Detection post-processing of the REAL reference: core/test.py:750-806 box_results_with_nms_and_limit (with the reference's
    compiled Cython NMS), utils/boxes.py:294-310 box_voting, and the Cython soft_nms (utils/cython_nms.pyx:98-203) in its three
    modes.  The boxes fed to box_results come from the reference's own bbox_transform / clip (utils/boxes.py)."""
    import utils.boxes as box_utils
    import core.test as ref_test
    nms_mod = sys.modules['utils.cython_nms']
    rs = np.random.RandomState(17)
    out = {}
    for name, (T, K, R, D, thr, nms_thr) in {'pp_boxes_k2': (1, 2, 600, 100, 0.05, 0.5), 'pp_boxes_k5': (1, 5, 400, 60, 0.05, 0.3),
                                              'pp_tubes_k2': (3, 2, 200, 30, 0.05, 0.5), 'pp_nolimit': (1, 3, 150, 0, 0.2, 0.5)}.items():
        H, W = 720, 1280
        xy = np.stack([rs.uniform(0, W - 60, R), rs.uniform(0, H - 60, R)], axis=1)
        wh = rs.uniform(8, 300, (R, 2))
        boxes = np.zeros((R, 4 * T), np.float32)
        for t in range(T):
            jit = rs.uniform(-4, 4, (R, 2))
            boxes[:, 4 * t:4 * t + 2] = xy + jit
            boxes[:, 4 * t + 2:4 * t + 4] = xy + jit + wh
        logits = rs.randn(R, K).astype(np.float32) * 2
        scores = (np.exp(logits) / np.exp(logits).sum(axis=1, keepdims=True)).astype(np.float32)
        deltas = (rs.randn(R, K * 4 * T) * np.tile([1.0, 1.0, 2.0, 2.0], K * T)).astype(np.float32)
        pred = box_utils.clip_tiled_boxes(box_utils.bbox_transform(boxes, deltas, (10., 10., 5., 5.)), (H, W, 3))
        cfg.MODEL.NUM_CLASSES, cfg.TEST.SCORE_THRESH, cfg.TEST.NMS, cfg.TEST.DETECTIONS_PER_IM = K, thr, nms_thr, D
        cfg.TEST.SOFT_NMS.ENABLED = cfg.TEST.BBOX_VOTE.ENABLED = False
        sc, bx, cls_boxes = ref_test.box_results_with_nms_and_limit(scores, pred)
        out[name + '_cfg'] = np.array([T, K, R, D, thr, nms_thr], dtype=np.float64)
        out[name + '_boxes'], out[name + '_scores'], out[name + '_deltas'], out[name + '_pred'] = boxes, scores, deltas, pred
        out[name + '_out_scores'], out[name + '_out_boxes'] = sc, bx
        out[name + '_out_counts'] = np.array([len(cls_boxes[j]) for j in range(1, K)], dtype=np.int64)
    # soft-NMS + box voting on one class worth of detections
    n = 300
    b = rs.uniform(0, 250, (n, 4)).astype(np.float32)
    b[:, 2:] = b[:, :2] + rs.uniform(5, 120, (n, 2)).astype(np.float32)
    dets = np.hstack((b, rs.uniform(0.01, 1, (n, 1)).astype(np.float32)))
    out['soft_dets'] = dets
    for method, code in (('hard', 0), ('linear', 1), ('gaussian', 2)):
        d, inds = nms_mod.soft_nms(np.ascontiguousarray(dets), np.float32(0.5), np.float32(0.3), np.float32(0.001), np.uint8(code))
        out['soft_%s_dets' % method], out['soft_%s_inds' % method] = np.asarray(d), np.asarray(inds, dtype=np.int64)
    keep = np.asarray(nms_mod.nms(dets, np.float32(0.5)))
    out['vote_top'] = dets[keep]
    out['vote_out'] = box_utils.box_voting(dets[keep], dets, 0.8)
    np.savez_compressed(os.path.join(HERE, 'reference_postproc.npz'), **out)
    print('wrote reference_postproc.npz', len(out), 'arrays')


if __name__ == '__main__':
    if '--only-postproc' in sys.argv:
        golden_postproc(_install_shims())
    elif '--only-tracker' in sys.argv:
        golden_tracker(_install_shims())
    elif '--only-clips' in sys.argv:
        golden_clips(_install_shims())
    elif '--only-weights' in sys.argv:
        golden_weights(_install_shims())
    elif '--only-lr' in sys.argv:
        golden_lr_policy(_install_shims())
    elif '--only-builders' in sys.argv:
        golden_builders(_install_shims())
    elif '--only-dataset' in sys.argv:
        golden_dataset(_install_shims())
    elif '--only-cfg' in sys.argv:
        _c = _install_shims()
        golden_cfg_defaults(_c)
        golden_cfg_files(_c)
    elif '--only-roi-data' in sys.argv:
        golden_roi_data(_install_shims())
    elif '--only-blob' in sys.argv:
        golden_blob(_install_shims())
    elif '--only-decode' in sys.argv:
        golden_decode(_install_shims())
    else:
        main()
        golden_postproc(sys.modules['core.config'].cfg)
        golden_tracker(sys.modules['core.config'].cfg)
        golden_blob(sys.modules['core.config'].cfg)
        golden_decode(sys.modules['core.config'].cfg)
        golden_builders(sys.modules['core.config'].cfg)
        golden_weights(sys.modules['core.config'].cfg)
        golden_clips(sys.modules['core.config'].cfg)
        golden_dataset(sys.modules['core.config'].cfg)
