#!/usr/bin/env python3
"""BASELINE config 5 end to end on one GPU, as one number: the product's own tools chained the way the reference
chains them (tools/test_net.py -> detections.pkl -> tools/compute_tracks.py; lib/core/test_engine.py:311-333,
lib/core/tracking_engine.py:758-795) on a VIDEO-shaped clip list -- V videos of F frames, one 8-frame clip around every frame
(stride-1 sliding window, border frames replicated: lib/utils/video.py:149-201) -- with host uint8 720 x 1280 frames:

    3D R-50 FPN3D keypoint R-CNN on the pipelined engine (uint8 upload, device pre-processing, hipGraph replay)  -> detections.pkl
    host Hungarian tracker (one core, sequential over videos: tracking_engine.py:689-694)                       -> detections_withTracks.pkl

Prints ONE JSON object: detector clips/s (whole run and steady state), tracker frames/s, their ratio.  Weights are synthetic
(random-init keeps ~100 detections per frame: TRACKING.CONF_FILTER_INITIAL_DETS is set to 0 so the tracker really matches 100 x 100
boxes per frame pair -- ~12x the persons of a PoseTrack frame, i.e. a pessimistic tracker load)."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

import _path  # noqa
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from detectandtrack_amd.utils.video import clip_frame_ids  # noqa: E402


def video_roidb(n_videos, n_frames, T, h=720, w=1280, seed=3):
    rs = np.random.RandomState(seed)
    roidb = []
    for v in range(n_videos):
        base = [rs.randint(0, 255, (h, w, 3)).astype(np.uint8) for _ in range(4)]
        video = [base[i % 4] if i < 4 else np.roll(base[i % 4], 7 * i, axis=1) for i in range(n_frames)]
        for k in range(n_frames):
            ids = clip_frame_ids(k, 0, n_frames - 1, T)           # (lib/utils/video.py:149-201: one clip per key frame, borders replicated)
            roidb.append({'image': [video[i] for i in ids], 'frame_ids': [('vid%04d' % v, i) for i in ids], 'height': h, 'width': w,
                          'name': 'images/vid%04d/%06d.jpg' % (v, k)})
    return roidb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='50')
    ap.add_argument('--videos', type=int, default=4)
    ap.add_argument('--frames', type=int, default=60)
    ap.add_argument('--clips-per-forward', type=int, default=4)
    ap.add_argument('--in-flight', type=int, default=3)
    ap.add_argument('--trunk-cache', type=int, default=64,
                    help='cfg.HIP.FRAME_TRUNK_CACHE: frames whose conv1 ... res2 output the pipelined engine keeps (a stride-1 clip list shares T - 1 of T frames '
                         'between consecutive clips: only new frames are uploaded and run through that prefix); 0 = every clip computed whole')
    a = ap.parse_args()
    import bench
    from detectandtrack_amd.core.config import cfg, cfg_from_cfg, assert_and_infer_cfg, reset_cfg
    from detectandtrack_amd.core import test_engine, tracking_engine
    T = 8
    c = bench.model_cfg(a.arch, T, 'bf16')
    c['HIP'].update(PIPELINE_DEPTH=a.in_flight, IMS_PER_FORWARD=a.clips_per_forward, CLIP_GRAPH=True, FRAME_TRUNK_CACHE=a.trunk_cache)
    c['TEST'].update(SCORE_THRESH=0.05)
    c['RNG_SEED'] = 3
    reset_cfg()
    cfg_from_cfg(c)
    assert_and_infer_cfg()
    cfg.TRACKING.CONF_FILTER_INITIAL_DETS = 0.0
    test_engine.SYNTHETIC_WEIGHTS = True
    roidb = video_roidb(a.videos, a.frames, T)
    out = tempfile.mkdtemp(prefix='dat_config5_')
    import torch
    t0 = time.perf_counter()
    test_engine.test_net_on_dataset(roidb, multi_gpu=False, output_dir=out)
    torch.cuda.synchronize()
    t_det = time.perf_counter() - t0
    st = test_engine.test_net.last_stats or {}
    json_data = [{'image': e['name'], 'height': e['height'], 'width': e['width']} for e in roidb]
    t1 = time.perf_counter()
    dets = tracking_engine.run_posetrack_tracking(out, json_data)
    t_trk = time.perf_counter() - t1
    n = len(roidb)
    ndet = float(np.mean([len(b) for b in dets['all_boxes'][1]]))
    tracks = dets['all_tracks'][1]
    assert len(tracks) == n and all(len(t) == len(b) for t, b in zip(tracks, dets['all_boxes'][1]))
    det_rate = st.get('steady_clips_per_s') or n / t_det
    res = {
        'workload': '3D R-%s FPN3D inference over %d videos x %d frames (one 8-frame 720x1280 clip per frame, stride 1, host uint8 frames) -> '
                    'detections.pkl -> Hungarian tracker -> detections_withTracks.pkl' % (a.arch, a.videos, a.frames),
        'clips': n, 'detector_seconds_incl_warmup': round(t_det, 3), 'detector_clips_per_s': round(det_rate, 2),
        'detector_clips_per_s_incl_warmup': round(n / t_det, 2),
        'engine': {'clips_per_forward': st.get('per_forward'), 'forwards_in_flight': st.get('in_flight'), 'hip_graph': st.get('hip_graph'),
                   'upload_mb_per_clip': round(st.get('upload_bytes_per_clip', 0) / 1e6, 2), 'host_path_images': st.get('host_path_images'), 'tie_rerun_images': st.get('tie_rerun_images'),
                   'frame_trunk_cache': st.get('frame_trunk_cache'),
                   'trunk_frames_computed': st.get('trunk_frames_computed'), 'trunk_frames_requested': st.get('trunk_frames_requested')},
        'detections_per_frame': round(ndet, 1),
        'tracker_seconds': round(t_trk, 3), 'tracker_frames_per_s': round(n / t_trk, 1), 'tracker_cores': 1,
        'tracker_over_detector': round((n / t_trk) / det_rate, 2),
        'end_to_end_clips_per_s': round(n / (n / det_rate + t_trk), 2),
    }
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
