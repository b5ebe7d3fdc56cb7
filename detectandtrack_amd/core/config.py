"""Global config `cfg` — same keys, defaults, precedence and helper names as the reference's
lib/core/config.py (defaults :37-760, assert_and_infer_cfg :764, _merge_a_into_b :787,
_fix_video_time_kernel_dim :839, cfg_from_file :865, cfg_from_list :880, get_output_dir :776).

Precedence: defaults < YAML file < `KEY VAL` command-line pairs (reference README.md:132).
The defaults live in one YAML document below (py3: all strings are `str`, not bytes).
"""
import copy
import os
import os.path as osp
from ast import literal_eval

import numpy as np
import yaml

from detectandtrack_amd.utils.collections import AttrDict

_DEFAULTS_YAML = r"""
TRAIN:
  WEIGHTS: ''
  DATASET: ''
  SCALES: [600]
  MAX_SIZE: 1000
  IMS_PER_BATCH: 2
  BATCH_SIZE_PER_IM: 64
  FG_FRACTION: 0.25
  FG_THRESH: 0.5
  BG_THRESH_HI: 0.5
  BG_THRESH_LO: 0.0
  USE_FLIPPED: true
  BBOX_REG: true
  BBOX_THRESH: 0.5
  SNAPSHOT_ITERS: 20000
  PROPOSAL_FILE: ''
  ASPECT_GROUPING: true
  RPN_POSITIVE_OVERLAP: 0.7
  RPN_NEGATIVE_OVERLAP: 0.3
  RPN_FG_FRACTION: 0.5
  RPN_BATCH_SIZE_PER_IM: 256
  RPN_NMS_THRESH: 0.7
  RPN_PRE_NMS_TOP_N: 12000
  RPN_POST_NMS_TOP_N: 2000
  RPN_STRADDLE_THRESH: 0
  RPN_MIN_SIZE: 0
  DROPOUT: 0.0
  CROWD_FILTER_THRESH: 0.7
  GT_MIN_AREA: -1
  MINIBATCH_QUEUE_SIZE: 64
  BBOX_NORMALIZE_TARGETS_PRECOMPUTED_deprecated: null
TEST:
  WEIGHTS: ''
  DATASET: ''
  DATASETS: []
  SCALES: [600]
  MAX_SIZE: 1000
  NMS: 0.3
  SOFT_NMS: {ENABLED: false, METHOD: linear, SIGMA: 0.5}
  BBOX_VOTE: {ENABLED: false, VOTE_TH: 0.8}
  SVM: false
  BBOX_REG: true
  PROPOSAL_FILE: ''
  PROPOSAL_FILES: []
  PROPOSAL_LIMIT: 2000
  RPN_NMS_THRESH: 0.7
  RPN_PRE_NMS_TOP_N: 12000
  RPN_POST_NMS_TOP_N: 2000
  RPN_MIN_SIZE: 0
  DETECTIONS_PER_IM: 100
  SCORE_THRESH: 0.05
  COMPETITION_MODE: true
  FORCE_JSON_DATASET_EVAL: false
  BBOX_AUG: {SCORE_HEUR: ID, COORD_HEUR: ID, H_FLIP: false, SCALES: [], MAX_SIZE: 4000, SCALE_H_FLIP: false,
             SCALE_SIZE_DEP: false, AREA_TH_LO: 2500, AREA_TH_HI: 32400, ASPECT_RATIOS: [], ASPECT_RATIO_H_FLIP: false}
  MASK_AUG: {HEUR: SOFT_AVG, H_FLIP: false, SCALES: [], MAX_SIZE: 4000, SCALE_H_FLIP: false, SCALE_SIZE_DEP: false,
             AREA_TH: 32400, ASPECT_RATIOS: [], ASPECT_RATIO_H_FLIP: false}
  KPS_AUG: {HEUR: HM_AVG, H_FLIP: false, SCALES: [], MAX_SIZE: 4000, SCALE_H_FLIP: false, SCALE_SIZE_DEP: false,
            AREA_TH: 32400, ASPECT_RATIOS: [], ASPECT_RATIO_H_FLIP: false}
  ENSEMBLE: {DEVSTORAGE_CACHE: false, RPN_CONFIGS: [], PROPOSAL_CACHE: /tmp}
  INIT_RANDOM_VARS_BEFORE_LOADING: false
  EXT_CNN_FEATURES: false
  EXT_CNN_FEATURES_MODEL: ImNet
MODEL:
  TYPE: ''
  CONV_BODY: ''
  ROI_HEAD: ''
  NUM_CLASSES: -1
  PS_GRID_SIZE: 3
  DILATION: 1
  CLS_AGNOSTIC_BBOX_REG: false
  RPN_ONLY: false
  FASTER_RCNN: false
  MASK_ON: false
  KEYPOINTS_ON: false
  EXECUTION_TYPE: dag
  BBOX_REG_WEIGHTS: [10.0, 10.0, 5.0, 5.0]
  VIDEO_ON: false
  USE_BN: false
  USE_BN_TESTMODE_ONLY: false
  BN_EPSILON: 1.0000001e-05
  BN_MOMENTUM: 0.9
SOLVER:
  BASE_LR: 0.001
  LR_POLICY: step
  GAMMA: 0.1
  STEP_SIZE: 30000
  MAX_ITER: 40000
  MOMENTUM: 0.9
  WEIGHT_DECAY: 0.0005
  WARM_UP_ITERS: 500
  WARM_UP_FACTOR: 0.3333333333333333
  WARM_UP_METHOD: linear
  STEPS: []
  LRS: []
  SCALE_MOMENTUM: true
  SCALE_MOMENTUM_THRESHOLD: 1.1
  LOG_LR_CHANGE_THRESHOLD: 1.1
FAST_RCNN: {MLP_HEAD_DIM: 1024, ROI_XFORM_METHOD: RoIPoolF, ROI_XFORM_SAMPLING_RATIO: 0, ROI_XFORM_RESOLUTION: 14}
RPN: {'ON': false, SIZES: [64, 128, 256, 512], STRIDE: 16, ASPECT_RATIOS: [0.5, 1, 2]}      # ('ON' quoted: YAML 1.1 reads a bare ON as the boolean true)
FPN:
  FPN_ON: false
  DIM: 256
  ZERO_INIT_LATERAL: false
  COARSEST_STRIDE: 32
  MULTILEVEL_ROIS: false
  ROI_CANONICAL_SCALE: 224
  ROI_CANONICAL_LEVEL: 4
  ROI_MAX_LEVEL: 5
  ROI_MIN_LEVEL: 2
  MULTILEVEL_RPN: false
  RPN_MAX_LEVEL: 6
  RPN_MIN_LEVEL: 2
  RPN_ASPECT_RATIOS: [0.5, 1, 2]
  RPN_ANCHOR_START_SIZE: 32
  EXTRA_CONV_LEVELS: false
  INPLACE_LATERAL: false
MRCNN: {MASK_HEAD_NAME: '', RESOLUTION: 14, ROI_XFORM_METHOD: RoIAlign, ROI_XFORM_RESOLUTION: 7,
        ROI_XFORM_SAMPLING_RATIO: 0, DIM_REDUCED: 256, THRESH_BINARIZE: 0.5, WEIGHT_LOSS_MASK: 1.0,
        CLS_SPECIFIC_MASK: true, DILATION: 2, UPSAMPLE_RATIO: 1, USE_FC_OUTPUT: false, CONV_INIT: GaussianFill}
KRCNN:
  ROI_KEYPOINTS_HEAD: ''
  HEATMAP_SIZE: -1
  UP_SCALE: -1
  USE_DECONV: false
  USE_DECONV_OUTPUT: false
  DILATION: 1
  DECONV_KERNEL: 4
  DECONV_DIM: 256
  NUM_KEYPOINTS: -1
  CONV_HEAD_DIM: 256
  CONV_HEAD_KERNEL: 3
  CONV_INIT: GaussianFill
  NMS_OKS: false
  KEYPOINT_CONFIDENCE: bbox
  ROI_XFORM_METHOD: RoIAlign
  ROI_XFORM_RESOLUTION: 7
  ROI_XFORM_SAMPLING_RATIO: 0
  MIN_KEYPOINT_COUNT_FOR_VALID_MINIBATCH: 20
  NUM_STACKED_CONVS: 8
  INFERENCE_MIN_SIZE: 0
  LOSS_WEIGHT: 1.0
  USE_3D_DECONV: false
  NO_3D_DECONV_TIME_TO_CH: false
VIDEO:
  NUM_FRAMES: -1
  NUM_FRAMES_MID: -1
  TIME_INTERVAL: -1
  WEIGHTS_INFLATE_MODE: ''
  TIME_KERNEL_DIM: {BODY: 1, HEAD_RPN: 1, HEAD_KPS: 1, HEAD_DET: 1}
  TIME_STRIDE_ON: false
  BODY_HEAD_LINK: ''
  PREDICT_RPN_BOX_VIS: false
  DEBUG_USE_RPN_GT: false
  RPN_TUBE_GEN_STYLE: replicate
  DEFAULT_CLIPS_PER_VIDEO: 9999999999
EXT_PATHS: {POSEVAL_CODE_PATH: ''}
RESNETS: {NUM_GROUPS: 1, WIDTH_PER_GROUP: 64, STRIDE_1X1: true, TRANS_FUNC: bottleneck_transformation}
TRACKING:
  CONF_FILTER_INITIAL_DETS: 0.9
  DETECTIONS_FILE: ''
  DISTANCE_METRICS: [bbox-overlap, cnn-cosdist, pose-pck]
  DISTANCE_METRIC_WTS: [1.0, 0.0, 0.0]
  BIPARTITE_MATCHING_ALGO: hungarian
  CNN_MATCHING_LAYER: layer3
  FLOW_SMOOTHING_ON: false
  KP_CONF_TYPE: global
  FLOW_SMOOTHING: {FLOW_SHOT_BOUNDARY_TH: 6.0, N_CONTEXT_FRAMES: 3, EXTEND_TRACKS: true}
  KEEP_CENTER_DETS_ONLY: true
  DEBUG: {UPPER_BOUND: false, UPPER_BOUND_2_GT_KPS: false, UPPER_BOUND_2_GT_KPS_ONLY_CONF: false,
          UPPER_BOUND_3_SHOTS: false, UPPER_BOUND_4_EVAL_UPPER_BOUND: false, UPPER_BOUND_5_GT_KPS_ONLY: false,
          FLOW_SMOOTHING_COMBINE: false, DUMMY_TRACKS: false}
  LSTM: {MODEL: LSTM, EMSIZE: 200, NHID: 200, NLAYERS: 2, DROPOUT: 0.2, TIED_WTS: false, LR: 0.1, GRAD_CLIP: 0.25,
         BATCH_SIZE: 20, EPOCHS: 10, LOG_INTERVAL: 200, LOSS_LAST_PRED_ONLY: false, FEATS_TO_CONSIDER: [bbox, kpts],
         NUM_WORKERS: 4, CONSIDER_SHORT_TRACKS_TOO: false}
  LSTM_TEST: {LSTM_TRACKING_ON: false, LSTM_WEIGHTS: ''}
EVAL: {EVAL_MPII_PER_VIDEO: false, EVAL_MPII_DROP_DETECTION_THRESHOLD: 0.5, EVAL_MPII_KPT_THRESHOLD: -.inf}
NUM_GPUS: 1
USE_NCCL: false
DEDUP_BOXES: 0.0625
RNG_SEED: 3
EPS: 1.0e-14
OUTPUT_DIR: /tmp
MATLAB: matlab
VOC_DIR: ''
ROOT_GPU_ID: 0
MEMONGER: true
MEMONGER_SHARE_ACTIVATIONS: false
VIS: false
VIS_THR: 0.9
FINAL_MSG: ''
ROIDB_SUBSET: []
NUM_WORKERS: 4
CLUSTER: {ON_CLUSTER: false, AUTO_RESUME: true}
DEVSTORAGE: {MOUNT_ENABLED: false, HOSTNAME: '', REMOTE_PATH: '', MOUNT_POINT: /tmp/devstorage}
DEBUG: {DATA_LOADING: false, STOP_TRAIN_ITER: false}
USE_GPU_NMS_deprecated: null
"""

# keys the reference stores as tuples (type-checked on merge); YAML gives lists
_TUPLE_KEYS = {'SCALES', 'DATASETS', 'PROPOSAL_FILES', 'ASPECT_RATIOS', 'RPN_CONFIGS', 'BBOX_REG_WEIGHTS', 'SIZES',
               'RPN_ASPECT_RATIOS', 'DISTANCE_METRICS', 'DISTANCE_METRIC_WTS'}


def _to_attr(d, key=None):
    if isinstance(d, dict):
        return AttrDict({k: _to_attr(v, k) for k, v in d.items()})
    if isinstance(d, list) and key in _TUPLE_KEYS:
        return tuple(d)
    return d


def _build_defaults():
    c = _to_attr(yaml.safe_load(_DEFAULTS_YAML))
    c.BBOX_XFORM_CLIP = np.log(1000. / 16.)                          # config.py:672
    c.PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])     # config.py:677 (BGR)
    c.ROOT_DIR = os.getcwd()
    # extension (not in the reference): arithmetic mode of the HIP path, 'bf16' (performance) | 'fp16' (round 6: IEEE-half operands at the
    # bf16 MFMA rate, libdat_hip_f16.so, process started with DAT_H16=fp16) | 'fp32' (parity: v_mfma_f32) |
    # 'bf16x3' (inference: fp32 activations, every conv on hi / lo bf16 splits of both operands -- x_hi*W_hi + x_hi*W_lo + x_lo*W_hi,
    # fp32 accumulate, ~2^-16 relative: the 1e-3 parity bar at several times the fp32-MFMA rate)
    # KEYFRAME_DCE (opt-in): with BODY_HEAD_LINK 'slice-center' the heads read only the centre frame of every FPN
    # output; True computes just that frame of convs whose output is consumed solely by SliceKeyFrame (identical
    # rois / scores / heatmaps, the unread frames of fpn_res*_sum are never materialised)
    # DEVICE_KPS_DECODE: heatmaps_to_keypoints (utils/keypoints.py:94-149) runs on the GPU; False = the reference's host loop
    # FRAME_TRUNK_CACHE (opt-in, frames kept): conv1 / pool1 / res2 have no temporal extent (ResNet3D.py:258-275, time kernel 1),
    # so their output for a FRAME does not depend on the clip around it; sliding-window inference (one clip per key frame,
    # stride 1, utils/video.py:149-201) re-uses it instead of recomputing (and re-uploading) T-1 of T frames per clip
    # DEVICE_BOX_RESULTS: the glue between model.net and model.keypoint_net (core/test.py:215-252 box decode, :750-806 score
    # threshold / per-class NMS / DETECTIONS_PER_IM) runs on the GPU -- no host synchronisation inside a clip; False = the host path
    # RCCL_DIRECT (training): the gradient all-reduce goes through the C ABI's dat_allreduce_bucket (RCCL called directly) instead of
    # torch.distributed.all_reduce; the process group is then only used to hand the communicator id to the ranks
    # FUSE_STEM_POOL: conv1 + affine + ReLU + pool1 (ResNet3D.py:258-265) as one kernel; the `conv1` blob is then not available
    # to FetchBlob (bit-identical pool1; False = two kernels)
    # PIPELINE_DEPTH: forwards in flight in core/test_engine.test_net (core/pipeline.ClipPipeline: own HIP stream + blob namespace per
    # slot, completion-order read-back, uint8 frame upload on a copy stream, dat_preprocess_frames); 1 = strictly sequential;
    # 0 = the reference's eager loop through im_detect_all (host pre-processing, fp32 upload).  CLIP_GRAPH: every slot replays its
    # forward as one captured hipGraph.  IMS_PER_FORWARD: independent images (2D models) / clips (3D models) per forward -- the N
    # axis of the blobs; every image keeps the results it gets alone (the reference runs one image per forward, core/test.py:212-214)
    # DEFER_WGRAD_FINISH (training): conv weight gradients accumulate in the kernels' own [tap][Cout][Cin] order in one flat buffer and ONE
    # launch per iteration turns them into gradients (training.py), instead of a memset + a finish launch around every layer's kernel.
    # MAX_GRAPHS_PER_SLOT: captured hipGraphs (one per input geometry, each with a private pool holding a forward's activations) kept
    # per pipeline slot, least recently used evicted.  PAD_TAIL_FORWARD: a last group of fewer than IMS_PER_FORWARD clips is padded
    # with repeats of its last clip (results dropped) instead of capturing a second graph for the smaller batch.
    # OVERLAP_ALLREDUCE (training, world size > 1): gradients are finished and all-reduced per bucket in reverse layer order on a
    # communication stream while the backward of the earlier layers continues (training.py); False = one exchange after the backward
    # FUSE_RELU_BWD (training): the ReLU backward of a blob with ONE reader is applied in the epilogue of that reader's data-gradient
    # conv (dat_conv3d_fwd res_mode 3) instead of a separate elementwise pass; identical gradients
    # PERSISTENT_CU_SHARE (pipelined inference engine, round 6): percent of the CUs the persistent HBM-bound conv kernels (one block per CU) take
    # while SEVERAL forwards are in flight (core/pipeline.py, depth >= 2): the other forwards' MFMA-bound kernels run beside them on the CUs
    # they leave (same box: R-50 forward +0.8 %, 2d_best +1.4 %, R-18 +0.3 %; a lone forward or a training iteration keeps 100)
    # FUSE_RELU_SUM_BWD (training, round 6): for a blob with TWO readers (a residual block's output) the data-gradient conv of the reader that
    # contributes last adds the other contribution AND applies the ReLU backward in its epilogue (dat_conv3d_fwd_sum_mask, res_mode 4)
    # WGRAD_PW_BATCH (training, with DEFER_WGRAD_FINISH): the weight gradients of up to this many POINTWISE convs (1 x 1 x 1) are queued and
    # run as one grouped launch (dat_conv3d_wgrad_acc_batch; also flushed whenever a gradient bucket completes); 0 = one launch per layer
    # DEVICE_ROI_SAMPLING (training): GenerateProposalLabels as one device kernel on the device-resident proposals (roi_data/device_sampler.py,
    # dat_sample_rois: the reference's candidate sets and counts, a counter-based draw instead of NumPy's stream); False = the host restatement
    # of lib/roi_data/fast_rcnn.py on a copy of the proposals (bit-compatible with the reference's numpy.random stream)
    # STEM_FROM_UINT8: on the host-frame path of the pipelined engine the fused stem (conv1 + affine + ReLU + pool1) reads the UPLOADED uint8
    # frames and evaluates prep_im_for_blob / im_list_to_blob (resize, mean, padding) in its patch loader (dat_stem_conv_pool_u8, bit-identical):
    # the fp32 `data` blob -- 99 MB per 720p clip -- is never written; False = dat_preprocess_frames writes it first
    # DECONV_GROUP_IGNORED: the keypoint deconv of a 3D head without KRCNN.NO_3D_DECONV_TIME_TO_CH is recorded with group = T
    # (model_builder.py:848-856).  False (default): a grouped ConvTranspose in Caffe2's filter layout (C_in, C_out / group, k, k) -- frame t
    # has its own [C, K, 4, 4] block.  True: what the pinned Caffe2 (b4e1588, Feb 2018: ConvTranspose has no `group` argument yet and brew
    # creates the full [T*C, T*K, 4, 4] filter) would execute -- the argument is dropped and the deconv is dense over T*C -> T*K channels
    c.HIP = AttrDict({'DTYPE': 'bf16', 'KEYFRAME_DCE': False, 'DEVICE_KPS_DECODE': True, 'FRAME_TRUNK_CACHE': 0,
                      'DEVICE_BOX_RESULTS': True, 'FUSE_STEM_POOL': True, 'RCCL_DIRECT': False,
                      'PIPELINE_DEPTH': 4, 'CLIP_GRAPH': True, 'IMS_PER_FORWARD': 1, 'FUSE_RELU_BWD': True, 'FUSE_RELU_SUM_BWD': True, 'PERSISTENT_CU_SHARE': 50, 'DET_SPARE_ROWS': 4,
                      'DEFER_WGRAD_FINISH': True, 'MAX_GRAPHS_PER_SLOT': 6, 'PAD_TAIL_FORWARD': True,
                      'OVERLAP_ALLREDUCE': True, 'WGRAD_PW_BATCH': 16, 'DEVICE_ROI_SAMPLING': True, 'DECONV_GROUP_IGNORED': False, 'STEM_FROM_UINT8': True})
    return c


__C = _build_defaults()
cfg = __C
cfg_default = copy.deepcopy(__C)


def reset_cfg():
    """Restore defaults in place (tests build several models in one process)."""
    fresh = copy.deepcopy(cfg_default)
    for k in list(__C.keys()):
        del __C[k]
    for k, v in fresh.items():
        __C[k] = v


def assert_and_infer_cfg():
    """config.py:764-773."""
    if __C.MODEL.RPN_ONLY or __C.MODEL.FASTER_RCNN:
        __C.RPN.ON = True
    if __C.MODEL.RPN_ONLY:
        __C.TRAIN.BBOX_REG = False
    if __C.VIDEO.NUM_FRAMES_MID == -1:
        __C.VIDEO.NUM_FRAMES_MID = __C.VIDEO.NUM_FRAMES
    assert (not __C.MODEL.USE_BN_TESTMODE_ONLY) or __C.MODEL.USE_BN


def get_output_dir(training=True):
    """config.py:776-784."""
    dataset = __C.TRAIN.DATASET if training else __C.TEST.DATASET
    outdir = osp.join(__C.OUTPUT_DIR, 'train' if training else 'test', dataset, __C.MODEL.TYPE)
    if not osp.exists(outdir):
        os.makedirs(outdir)
    return outdir


def _coerce(old, new, key):
    """Type rule of config.py:810-822: the new value must have the default's type."""
    if new is None or old is None or type(old) is type(new):
        return new
    if isinstance(old, np.ndarray):
        return np.array(new, dtype=old.dtype)
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(old), type(new), key))
    raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(old), type(new), key))


def _merge_a_into_b(a, b):
    """config.py:787-836."""
    if not isinstance(a, dict):
        return
    for k, v in a.items():
        if k not in b:
            if k + '_deprecated' in b:
                continue
            raise KeyError('{} is not a valid config key'.format(k))
        if isinstance(v, dict):
            if not isinstance(b[k], dict):
                raise ValueError('Type mismatch (dict vs. {}) for config key: {}'.format(type(b[k]), k))
            _merge_a_into_b(v, b[k])
            continue
        if isinstance(v, str):
            try:
                v = literal_eval(v)
            except BaseException:
                pass
        b[k] = _coerce(b[k], v, k)


def _fix_video_time_kernel_dim(a):
    """config.py:839-850: `VIDEO.TIME_KERNEL_DIM: 3` means every sub-key = 3."""
    if 'VIDEO' in a and 'TIME_KERNEL_DIM' in a['VIDEO'] and isinstance(a['VIDEO']['TIME_KERNEL_DIM'], int):
        val = a['VIDEO']['TIME_KERNEL_DIM']
        a['VIDEO']['TIME_KERNEL_DIM'] = {k: val for k in __C.VIDEO.TIME_KERNEL_DIM.keys()}
    return a


def cfg_from_file(filename):
    """config.py:865-872."""
    with open(filename, 'r') as f:
        yaml_cfg = yaml.safe_load(f)
    cfg_from_cfg(yaml_cfg)


def cfg_from_cfg(yaml_cfg):
    _merge_a_into_b(_fix_video_time_kernel_dim(dict(yaml_cfg)), __C)


def cfg_from_list(cfg_list):
    """config.py:880-900: KEY VAL pairs, e.g. ['TEST.WEIGHTS', 'x.pkl', 'NUM_GPUS', '1']."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        d = __C
        parts = k.split('.')
        for sub in parts[:-1]:
            assert sub in d, 'Config key {} not found'.format(sub)
            d = d[sub]
        sub = parts[-1]
        assert sub in d, 'Config key {} not found'.format(sub)
        try:
            value = literal_eval(v) if isinstance(v, str) else v
        except BaseException:
            value = v
        if isinstance(d[sub], tuple) and isinstance(value, list):
            value = tuple(value)
        assert d[sub] is None or isinstance(value, type(d[sub])), \
            'type {} does not match original type {}'.format(type(value), type(d[sub]))
        d[sub] = value
