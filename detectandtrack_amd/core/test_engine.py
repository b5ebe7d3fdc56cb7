"""Dataset-level inference engine — mirror of reference lib/core/test_engine.py (model init :50-74, the clip loop
:124-204, per-GPU fan-out :278-308, detections.pkl layout :199-204).

The clip list is a "roidb": a list of dicts with `image` (list of T frame paths or arrays).  It comes from cfg.TEST.DATASET through
`get_roidb_and_dataset` (datasets/json_dataset.py: a pycocotools-free reader of the COCO-format lists; utils/video.get_clip), or -- the
PoseTrack files are not available offline -- from a pickled / synthetic clip list handed in by the tools.
"""
import logging
import os
import pickle
from collections import defaultdict

import numpy as np
import yaml

from detectandtrack_amd.core.config import cfg, get_output_dir
from detectandtrack_amd.core.test import im_detect_all
from detectandtrack_amd.modeling import model_builder
from detectandtrack_amd.utils.timer import Timer
from detectandtrack_amd.utils import dist as dist_utils
from detectandtrack_amd.utils import net as net_utils
from detectandtrack_amd import workspace

logger = logging.getLogger(__name__)


SYNTHETIC_WEIGHTS = False     # tools/test_net.py --synthetic-weights: utils.net.synthetic_params instead of the builder's init


def initialize_model_from_cfg():
    """(:50-74) build the inference model, load TEST.WEIGHTS (or random-init when empty), create the nets."""
    model = model_builder.create(cfg.MODEL.TYPE, train=False)
    ws = workspace.GlobalWorkspace()
    if SYNTHETIC_WEIGHTS and not cfg.TEST.WEIGHTS:
        for k, v in net_utils.synthetic_params(model, cfg.RNG_SEED).items():
            ws.set_param(k, v)
    else:
        net_utils.initialize_params(model, ws)
    if cfg.TEST.WEIGHTS:
        net_utils.initialize_from_weights_file(model, ws, cfg.TEST.WEIGHTS)
    ws.CreateNet(model.net)
    ws.CreateNet(model.conv_body_net)
    if cfg.MODEL.KEYPOINTS_ON:
        ws.CreateNet(model.keypoint_net)
    return model


def get_roidb_and_dataset(ind_range, include_gt=False):
    """(:77-103) the clip list of cfg.TEST.DATASET: the COCO-format annotation file -> roidb (datasets/json_dataset.py), video models ->
    one clip per key frame (utils/video.get_clip).  Entries whose frames are files carry their paths as `frame_ids`, so the per-frame trunk
    cache (cfg.HIP.FRAME_TRUNK_CACHE) recognises the frames that consecutive clips share."""
    from detectandtrack_amd.datasets.json_dataset import JsonDataset
    from detectandtrack_amd.utils import video as video_utils
    dataset = JsonDataset(cfg.TEST.DATASET)
    if cfg.MODEL.FASTER_RCNN:
        roidb = dataset.get_roidb(gt=include_gt)
    else:
        roidb = dataset.get_roidb(gt=include_gt, proposal_file=cfg.TEST.PROPOSAL_FILE, proposal_limit=cfg.TEST.PROPOSAL_LIMIT)
    if cfg.MODEL.VIDEO_ON:
        roidb = video_utils.get_clip(roidb, remove_imperfect=False)
        for e in roidb:
            if all(isinstance(p, str) for p in e['image']):
                e['frame_ids'] = list(e['image'])
    total = len(roidb)
    start, end = (0, total) if ind_range is None else ind_range
    return roidb[start:end] if ind_range is not None else roidb, dataset, start, end, total


def empty_results(num_classes, num_images):
    """(:258-275)"""
    all_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    all_keyps = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    all_segms = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    return all_boxes, all_segms, all_keyps


def extend_results(index, all_res, im_res):
    for cls_idx in range(1, len(im_res)):
        all_res[cls_idx][index] = im_res[cls_idx]


_FRAME_CACHE = None      # path -> decoded BGR uint8 frame, least recently used first (sliding-window clips share T - 1 of T frames)


def read_frame(path, cache_frames=64):
    """One image file -> HxWx3 uint8 BGR (the layout cv2.imread hands the reference, lib/utils/video.py / datasets roidb
    `image` paths, lib/core/test_engine.py:124-204).  Decoded with Pillow (OpenCV is not in the image); an LRU of decoded frames
    makes a stride-1 sliding window over a video decode every file once.  The EXIF orientation is applied like cv2.imread does
    (IMREAD_COLOR honours it since OpenCV 3.1); Pillow's JPEG IDCT / chroma upsampling may still differ from libjpeg-turbo-in-OpenCV by
    +-1 LSB -- pixel parity on real JPEGs is not claimed.  The cached arrays are shared by every clip that contains the frame and are
    therefore READ-ONLY: a consumer that flips or augments must copy."""
    global _FRAME_CACHE
    import collections
    if _FRAME_CACHE is None:
        _FRAME_CACHE = collections.OrderedDict()
    hit = _FRAME_CACHE.get(path)
    if hit is not None:
        _FRAME_CACHE.move_to_end(path)
        return hit
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        rgb = np.asarray(ImageOps.exif_transpose(im).convert('RGB'), dtype=np.uint8)
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])
    bgr.setflags(write=False)
    _FRAME_CACHE[path] = bgr
    while len(_FRAME_CACHE) > cache_frames:
        _FRAME_CACHE.popitem(last=False)
    return bgr


def load_clip(entry):
    """roidb entry -> list of T BGR frames: arrays are used as they are, paths are decoded (read_frame)."""
    frames = entry['image'] if isinstance(entry['image'], (list, tuple)) else [entry['image']]
    return [f if isinstance(f, np.ndarray) else read_frame(str(f)) for f in frames]


def _pipelined(part):
    """Can this clip list run on the pipelined engine?  It needs the device post-processing path and frames given as uint8 arrays; with
    cfg.HIP.FRAME_TRUNK_CACHE > 0 also frame ids on every entry (round 5: core/pipeline.FrameTrunkCache)."""
    from detectandtrack_amd.core import test as engine
    if int(cfg.HIP.get('PIPELINE_DEPTH', 0)) < 1 or not engine.device_results_supported() or cfg.MODEL.MASK_ON:
        return False
    if cfg.TEST.COMPETITION_MODE or cfg.HIP.KEYFRAME_DCE:
        return False
    if cfg.HIP.FRAME_TRUNK_CACHE > 0 and not (cfg.MODEL.VIDEO_ON and all('frame_ids' in e for e in part)):
        return False        # (the pipelined engine's per-frame trunk cache needs frame ids on every entry; otherwise: the eager loop)
    first = load_clip(part[0])          # (frame dtypes are checked clip by clip inside the loop: a non-uint8 clip takes the eager path)
    if any(f.dtype != np.uint8 for f in first):
        return False
    return True


def _test_net_pipelined(model, part, all_boxes, all_keyps, timers):
    """The clip loop on core/pipeline.ClipPipeline: cfg.HIP.IMS_PER_FORWARD clips of equal frame size per forward,
    cfg.HIP.PIPELINE_DEPTH forwards in flight, results placed by clip index (the detections.pkl order never depends on the
    completion order)."""
    from detectandtrack_amd.core.pipeline import ClipPipeline
    pipe = ClipPipeline(model, workspace.GlobalWorkspace(), depth=int(cfg.HIP.PIPELINE_DEPTH), graph=bool(cfg.HIP.CLIP_GRAPH))
    per = max(1, int(cfg.HIP.IMS_PER_FORWARD))

    def place(done):
        for tags, res in done:
            for i, (cls_boxes_i, _, cls_keyps_i) in zip(tags, res):
                if i is None:           # a padding clip of a tail group
                    continue
                extend_results(i, all_boxes, cls_boxes_i)
                if cls_keyps_i is not None:
                    extend_results(i, all_keyps, cls_keyps_i)

    def submit(group):
        clips, tags = [c for _, c in group], [j for j, _ in group]
        fids = [part[j].get('frame_ids') for j in tags]
        fids = fids if (cfg.HIP.FRAME_TRUNK_CACHE > 0 and all(f is not None for f in fids)) else None
        if pipe.use_graph and cfg.HIP.get('PAD_TAIL_FORWARD', True) and len(clips) < per and per in batches_seen.get(shape_of(clips[0]), ()):
            # a short group of a geometry whose full-size graph exists: repeat the last clip (results dropped) instead of capturing a
            # second graph -- with its own pool of activations -- for the smaller batch
            pad = per - len(clips)
            clips, tags = clips + [clips[-1]] * pad, tags + [None] * pad
            fids = fids + [fids[-1]] * pad if fids is not None else None
        batches_seen.setdefault(shape_of(clips[0]), set()).add(len(clips))
        pipe.submit_frames(clips, tag=tags, frame_ids=fids)

    def shape_of(clip):
        return (len(clip),) + tuple(clip[0].shape)

    def eager(i, clip, entry):
        """a clip the pipelined engine does not take (frames that are not uint8): the reference loop, with nothing else in flight"""
        place(pipe.drain())
        cls_boxes_i, _, cls_keyps_i = im_detect_all(model, clip, None, timers, frame_ids=entry.get('frame_ids'))
        extend_results(i, all_boxes, cls_boxes_i)
        if cls_keyps_i is not None:
            extend_results(i, all_keyps, cls_keyps_i)
    batches_seen = {}
    timers['im_detect_bbox'].tic()
    group, gshape = [], None
    for i, entry in enumerate(part):
        clip = load_clip(entry)
        if any(f.dtype != np.uint8 for f in clip):
            if group:
                submit(group)
                group = []
            eager(i, clip, entry)
            continue
        shape = shape_of(clip)
        if group and (shape != gshape or len(group) == per):
            submit(group)
            group = []
        group.append((i, clip))
        gshape = shape
        if len(pipe.results) > 64:
            done, pipe.results = pipe.results, []
            place(done)
    if group:
        submit(group)
    place(pipe.drain())
    timers['im_detect_bbox'].toc()
    return pipe


def test_net(roidb, ind_range=None, output_dir=None):
    """Run the detector over roidb[start:end] and dump detection_range_s_e.pkl (:124-204).  Execution mode: the pipelined engine
    (core/pipeline.py; cfg.HIP.PIPELINE_DEPTH >= 1, default) or the reference's one-clip-at-a-time loop through im_detect_all."""
    model = initialize_model_from_cfg()
    start, end = (0, len(roidb)) if ind_range is None else ind_range
    part = roidb[start:end]
    num_classes = cfg.MODEL.NUM_CLASSES
    all_boxes, all_segms, all_keyps = empty_results(num_classes, len(part))
    timers = defaultdict(Timer)
    test_net.last_stats = None
    if part and _pipelined(part):
        pipe = _test_net_pipelined(model, part, all_boxes, all_keyps, timers)
        rate = pipe.rate()
        test_net.last_stats = {'clips': len(part), 'seconds': timers['im_detect_bbox'].total_time, 'steady_clips_per_s': rate,
                               'upload_bytes_per_clip': pipe.upload_bytes / float(len(part)), 'host_submit_ms_per_clip': 1e3 * pipe.host_enqueue_s / len(part),
                               'host_staging_ms_per_clip': 1e3 * pipe.stage_s / len(part), 'host_path_images': pipe.host_path_images, 'tie_rerun_images': pipe.rerun_images, 'in_flight': int(cfg.HIP.PIPELINE_DEPTH),
                               'per_forward': int(cfg.HIP.IMS_PER_FORWARD), 'hip_graph': bool(cfg.HIP.CLIP_GRAPH),
                               'frame_trunk_cache': int(pipe.trunk.capacity) if pipe.trunk is not None else 0,
                               'trunk_resets': pipe.trunk.resets if pipe.trunk is not None else None,
                               'trunk_frames_computed': pipe.trunk.frames_computed if pipe.trunk is not None else None,
                               'trunk_frames_requested': pipe.trunk.frames_requested if pipe.trunk is not None else None}
        logger.info('im_detect: range [%d, %d] of %d: %d clips in %.3fs incl. warm-up%s (pipelined: %d in flight, %d per forward, '
                    'hipGraph %s, %.1f MB uploaded per clip)', start + 1, end, len(roidb), len(part), timers['im_detect_bbox'].total_time,
                    ', steady state %.1f clips/s' % rate if rate else '', int(cfg.HIP.PIPELINE_DEPTH), int(cfg.HIP.IMS_PER_FORWARD),
                    bool(cfg.HIP.CLIP_GRAPH), pipe.upload_bytes / 1e6 / len(part))
    else:
        for i, entry in enumerate(part):
            cls_boxes_i, cls_segms_i, cls_keyps_i = im_detect_all(model, load_clip(entry), None, timers, frame_ids=entry.get('frame_ids'))
            extend_results(i, all_boxes, cls_boxes_i)
            if cls_keyps_i is not None:
                extend_results(i, all_keyps, cls_keyps_i)
            if i % 10 == 0:
                logger.info('im_detect: range [%d, %d] of %d: %d/%d  bbox %.3fs misc_bbox %.3fs kps %.3fs misc_kps %.3fs',
                            start + 1, end, len(roidb), i + 1, len(part), timers['im_detect_bbox'].average_time,
                            timers['misc_bbox'].average_time, timers['im_detect_keypoints'].average_time,
                            timers['misc_keypoints'].average_time)
    res = dict(all_boxes=all_boxes, all_segms=all_segms, all_keyps=all_keyps, cfg=yaml.safe_dump(net_utils._plain(cfg)))   # (:199-204)
    if output_dir is not None:
        name = 'detection_range_%s_%s.pkl' % (start, end) if ind_range is not None else 'detections.pkl'
        with open(os.path.join(output_dir, name), 'wb') as f:
            pickle.dump(res, f, pickle.HIGHEST_PROTOCOL)
    return res


def merge_range_results(parts):
    """Concatenate per-range results in range order (:286-297)."""
    merged = None
    for p in parts:
        if merged is None:
            merged = {k: (v if k == 'cfg' else [list(c) for c in v]) for k, v in p.items()}
            continue
        for k in merged:
            if k == 'cfg':
                continue
            for cls in range(len(merged[k])):
                merged[k][cls] += p[k][cls]
    return merged


def test_net_on_dataset(roidb, multi_gpu=False, output_dir=None):
    """(:311-333) under torch.distributed each rank takes its contiguous clip range; rank 0 merges and writes
    detections.pkl.  Without a launcher it is the single-GPU path."""
    output_dir = output_dir or get_output_dir(training=False)
    dist = dist_utils.init_process_group() if multi_gpu else None
    if dist is None:
        return test_net(roidb, None, output_dir)
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = dist_utils.shard_range(len(roidb), world, rank)
    local = test_net(roidb, rng, output_dir)
    parts = dist_utils.gather_in_range_order([local], dist)
    if rank != 0:
        return None
    merged = merge_range_results(parts)
    with open(os.path.join(output_dir, 'detections.pkl'), 'wb') as f:
        pickle.dump(merged, f, pickle.HIGHEST_PROTOCOL)
    return merged
