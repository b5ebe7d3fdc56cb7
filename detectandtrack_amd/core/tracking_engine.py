"""Host tracker — mirror of reference lib/core/tracking_engine.py (the part every shipped config runs).

Stays on the host by design (BASELINE.json north_star): per video, frame t's detections are matched to frame t-1's
with a cost 1 - IoU (reference :158-181, Cython IoU lib/utils/cython_bbox.pyx) by the Hungarian algorithm
(`scipy.optimize.linear_sum_assignment`, :237) or greedily (:184-206); unmatched detections open new track ids
modulo MAX_TRACK_IDS (:339-348).  Before that: keep the centre frame of each tube (:751-755) and prune low-confidence
/ tiny boxes (:731-748).  Detections file: `{'all_boxes': [[], per-image n x (4T+1)], 'all_keyps': [[], per-image
list of 4 x 17T], ...}`; output adds `all_tracks` (:706).

Not mirrored (unused by the shipped configs / missing upstream): cnn-cosdist and pose-pck costs (weights 0.0),
LSTM tracker (module absent in the reference), optical-flow smoothing (needs OpenCV), debug upper bounds.
"""
import logging
import os.path as osp
import pickle
import time

import numpy as np
import scipy.optimize

from detectandtrack_amd.core.config import cfg
import detectandtrack_amd.utils.boxes as box_utils

logger = logging.getLogger(__name__)

MAX_TRACK_IDS = 999
FIRST_TRACK_ID = 0


def _load_det_file(path):
    with open(path, 'rb') as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:
            f.seek(0)
            return pickle.load(f, encoding='latin1')


def _write_det_file(dets, path):
    with open(path, 'wb') as f:
        pickle.dump(dets, f, pickle.HIGHEST_PROTOCOL)


def _image_path(entry):
    """utils/image.py:44-48: a clip is named by its CENTRE frame (the key frame), a single image by itself -- the key the frames of a video
    are sorted by (:681) and the directory a video is told apart by (:65-67)."""
    names = entry['image']
    return names[len(names) // 2] if isinstance(names, (list, tuple)) else names


def _is_same_video(a, b):
    return osp.dirname(_image_path(a)) == osp.dirname(_image_path(b))


def _get_boxes(dets, i):
    return dets['all_boxes'][1][i]


def _get_poses(dets, i):
    return dets['all_keyps'][1][i]


def _center_boxes(boxes):
    """n x (4T+1) -> n x 5: the centre frame's box + score (:83-91)."""
    if len(boxes) == 0:
        return boxes
    assert (boxes.shape[-1] - 1) % 4 == 0, 'Must contain scores in last col.'
    c = ((boxes.shape[-1] - 1) // 4) // 2
    return boxes[:, list(range(c * 4, (c + 1) * 4)) + [-1]]


def _center_poses(poses):
    if len(poses) == 0:
        return poses
    K = cfg.KRCNN.NUM_KEYPOINTS
    c = (poses[0].shape[-1] // K) // 2
    return [p[..., c * K:(c + 1) * K] for p in poses]


def _center_detections(dets):
    for i in range(len(dets['all_boxes'][1])):
        dets['all_boxes'][1][i] = _center_boxes(_get_boxes(dets, i))
        dets['all_keyps'][1][i] = _center_poses(_get_poses(dets, i))


def _prune_bad_detections(dets, json_data, conf):
    """score >= conf and (clipped) area >= 50 (:711-748)."""
    for i in range(len(dets['all_boxes'][1])):
        boxes, poses = dets['all_boxes'][1][i], dets['all_keyps'][1][i]
        if len(boxes) == 0:
            continue
        ht, wd = json_data[i]['height'], json_data[i]['width']
        boxes[:, 0] = np.maximum(boxes[:, 0], 0)
        boxes[:, 1] = np.maximum(boxes[:, 1], 0)
        boxes[:, 2] = np.minimum(boxes[:, 2], wd)
        boxes[:, 3] = np.minimum(boxes[:, 3], ht)
        big = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) >= 50
        sel = np.where(np.logical_and(boxes[:, -1] >= conf, big))[0]
        dets['all_boxes'][1][i] = boxes[sel]
        dets['all_keyps'][1][i] = [poses[j] for j in sel.tolist()]
    return dets


def _compute_pairwise_iou(a, b):
    return box_utils.bbox_overlaps(a, b)


def _compute_distance_matrix(prev_boxes, cur_boxes, cost_types, cost_weights):
    """Weighted sum of the enabled pairwise costs (:158-181)."""
    assert len(cost_weights) == len(cost_types)
    costs = []
    for kind, wt in zip(cost_types, cost_weights):
        if wt == 0:
            continue
        if kind == 'bbox-overlap':
            costs.append((1 - _compute_pairwise_iou(prev_boxes, cur_boxes)) * wt)
        else:
            raise NotImplementedError('cost type {} (weight 0.0 in every shipped config)'.format(kind))
    return np.sum(np.stack(costs, axis=0), axis=0)


def bipartite_matching_greedy(C):
    """Repeatedly take the globally cheapest (row, col) pair (:184-206)."""
    C = C.copy()
    rows, cols = np.arange(C.shape[0]), np.arange(C.shape[1])
    prev_ids, cur_ids = [], []
    while C.size > 0:
        i, j = np.unravel_index(C.argmin(), C.shape)
        prev_ids.append(rows[i])
        cur_ids.append(cols[j])
        C = np.delete(np.delete(C, i, 0), j, 1)
        rows, cols = np.delete(rows, i), np.delete(cols, j)
    return prev_ids, cur_ids


def _compute_matches(prev_boxes, cur_boxes, cost_types, cost_weights, algo, C=None):
    """For every current box the index of the previous-frame box it continues, or -1 (:209-246)."""
    if C is None:
        matches = -np.ones((cur_boxes.shape[0],), dtype=np.int32)
        C = _compute_distance_matrix(prev_boxes, cur_boxes, cost_types, cost_weights)
    else:
        matches = -np.ones((C.shape[1],), dtype=np.int32)
    if algo == 'hungarian':
        prev_inds, next_inds = scipy.optimize.linear_sum_assignment(C)
    elif algo == 'greedy':
        prev_inds, next_inds = bipartite_matching_greedy(C)
    else:
        raise NotImplementedError('Unknown matching algo: {}'.format(algo))
    for p, n in zip(prev_inds, next_inds):
        matches[n] = p
    return matches


def _compute_tracks_video(video_json_data, dets):
    """Track ids for each frame of one video (:272-350)."""
    video_tracks = []
    next_id = FIRST_TRACK_ID
    for frame_id in range(len(video_json_data)):
        _, det_id = video_json_data[frame_id]
        cur_boxes = _get_boxes(dets, det_id)
        if frame_id == 0:
            matches = -np.ones((cur_boxes.shape[0],))
        else:
            prev_boxes = _get_boxes(dets, video_json_data[frame_id - 1][1])
            matches = _compute_matches(prev_boxes, cur_boxes, cfg.TRACKING.DISTANCE_METRICS,
                                       cfg.TRACKING.DISTANCE_METRIC_WTS, cfg.TRACKING.BIPARTITE_MATCHING_ALGO)
        prev_tracks = video_tracks[frame_id - 1] if frame_id > 0 else None
        frame_tracks = []
        for m in matches:
            if m == -1:
                frame_tracks.append(next_id)
                next_id += 1
                if next_id >= MAX_TRACK_IDS:
                    next_id %= MAX_TRACK_IDS
            else:
                frame_tracks.append(prev_tracks[int(m)])
        video_tracks.append(frame_tracks)
    return video_tracks


def split_into_videos(json_data):
    """Consecutive roidb entries whose images share a directory form one video (:669-685)."""
    videos, cur = [], []
    for i in range(len(json_data)):
        if i == 0 or _is_same_video(json_data[i - 1], json_data[i]):
            cur.append((json_data[i], i))
        else:
            videos.append(sorted(cur, key=lambda x: _image_path(x[0])))
            cur = [(json_data[i], i)]
    if cur:
        videos.append(cur)
    return videos


def compute_matches_tracks(json_data, dets, lstm_model=None):
    """(:669-708) videos are processed sequentially on one core, as in the reference (:689-694)."""
    if cfg.TRACKING.LSTM_TEST.LSTM_TRACKING_ON or cfg.TRACKING.FLOW_SMOOTHING_ON:
        raise NotImplementedError('LSTM tracking / flow smoothing are off in every shipped config')
    all_tracks = [[]] * len(json_data)
    videos = split_into_videos(json_data)
    assert len(json_data) == sum(len(v) for v in videos)
    for video in videos:
        tracks = _compute_tracks_video(video, dets)
        for i, (_, det_id) in enumerate(video):
            all_tracks[det_id] = tracks[i]
    dets['all_tracks'] = [[], all_tracks]
    return dets


def run_posetrack_tracking(test_output_dir, json_data):
    """(:758-795) detections.pkl -> detections_withTracks.pkl.  PoseTrack evaluation (poseval / MOTA) needs the
    dataset annotations, which are not available offline; it is out of the hot-path scope."""
    det_file = cfg.TRACKING.DETECTIONS_FILE if len(cfg.TRACKING.DETECTIONS_FILE) else \
        osp.join(test_output_dir, 'detections.pkl')
    out_file = osp.join(test_output_dir, 'detections_withTracks.pkl')
    if not osp.exists(det_file):
        raise ValueError('Output file not found {}'.format(det_file))
    dets = _load_det_file(det_file)
    if cfg.TRACKING.KEEP_CENTER_DETS_ONLY:
        _center_detections(dets)
    assert len(json_data) == len(dets['all_boxes'][1]) == len(dets['all_keyps'][1])
    dets = _prune_bad_detections(dets, json_data, cfg.TRACKING.CONF_FILTER_INITIAL_DETS)
    dets = compute_matches_tracks(json_data, dets)
    _write_det_file(dets, out_file)
    return dets


# ---- synthetic workload + timing (the CPU baseline reported next to the GPU numbers) -----------------
def synthetic_detections(n_videos=50, n_frames=100, n_persons=8, seed=3, T=1):
    """Random-walk person boxes with detector-like scores; returns (json_data, dets)."""
    rs = np.random.RandomState(seed)
    json_data, boxes_all, keyps_all = [], [], []
    for v in range(n_videos):
        n = max(1, n_persons + rs.randint(-2, 3))
        ctr = rs.uniform([100, 100], [1180, 620], (n, 2))
        size = rs.uniform(60, 220, (n, 2))
        for f in range(n_frames):
            ctr += rs.randn(n, 2) * 6
            vis = rs.uniform(size=n) > 0.08
            b = np.hstack((ctr - size / 2, ctr + size / 2))[vis]
            b = b + rs.randn(*b.shape) * 2
            sc = rs.uniform(0.90, 1.0, (b.shape[0], 1))
            tube = np.hstack([b + rs.randn(*b.shape) for _ in range(T)] + [sc]).astype(np.float32)
            boxes_all.append(tube)
            keyps_all.append([np.zeros((4, 17 * T), np.float32) for _ in range(tube.shape[0])])
            json_data.append({'image': 'images/vid%04d/%06d.jpg' % (v, f), 'height': 720, 'width': 1280})
    return json_data, {'all_boxes': [[], boxes_all], 'all_keyps': [[], keyps_all]}


def benchmark_tracking(json_data, dets):
    """Time the tracking path proper (centre-frame selection, pruning, per-video matching) on prepared detections."""
    t0 = time.time()
    _center_detections(dets)
    dets = _prune_bad_detections(dets, json_data, cfg.TRACKING.CONF_FILTER_INITIAL_DETS)
    dets = compute_matches_tracks(json_data, dets)
    return {'seconds': time.time() - t0, 'frames': len(json_data)}


def benchmark_synthetic(n_videos=50, n_frames=100, n_persons=8, seed=3):
    json_data, dets = synthetic_detections(n_videos, n_frames, n_persons, seed)
    r = benchmark_tracking(json_data, dets)
    el, n = r['seconds'], r['frames']
    return {'value': n / el, 'unit': 'frames/s', 'cores': 1, 'kind': 'port', 'seconds': el,
            'sample': '%d videos x %d frames x ~%d persons (synthetic detections, seed %d), Hungarian matching, '
                      'sequential over videos as reference tracking_engine.py:689-694' % (n_videos, n_frames, n_persons, seed)}


def _bench_worker_main(argv):
    """`python -m detectandtrack_amd.core.tracking_engine --bench-worker SEED N_VIDEOS N_FRAMES`: one persistent worker of bench.py's all-cores
    tracker leg.  Builds its videos' synthetic detections, prints 'ready', waits for a line on stdin, tracks, prints '<seconds> <frames>'."""
    import sys
    seed, n_videos, n_frames = int(argv[0]), int(argv[1]), int(argv[2])
    json_data, dets = synthetic_detections(n_videos, n_frames, 8, seed)
    benchmark_tracking(*synthetic_detections(1, 20, 8, seed))            # (first-call costs outside the clock)
    sys.stdout.write('ready\n')
    sys.stdout.flush()
    if not sys.stdin.readline():
        return
    r = benchmark_tracking(json_data, dets)
    sys.stdout.write('%.6f %d\n' % (r['seconds'], r['frames']))
    sys.stdout.flush()


if __name__ == '__main__':
    import sys
    if len(sys.argv) >= 5 and sys.argv[1] == '--bench-worker':
        _bench_worker_main(sys.argv[2:5])
