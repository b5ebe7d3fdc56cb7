"""PoseTrack prediction export — the wire format poseval reads (reference lib/core/mpii_eval_engine.py:30-181, 236-285;
SURVEY.md §8 f-3).  The evaluation itself (poseval, ground-truth .mat/.json files) needs the dataset, which is not available
offline; this module writes exactly the per-video `{'annolist': [...]}` JSON the reference hands to poseval, so MOTA / mAP can
be produced wherever the data is mounted.

Per detection the 17 network keypoints (PoseTrack-trained order `posetrack_src_keypoints`) are re-ordered to the 15 MPII-style
`dst_keypoints` ('neck' = mid-shoulder and 'head_top' = nose reflected about it when the source lacks them), thresholded by
EVAL.EVAL_MPII_KPT_THRESHOLD on the keypoint logit, and scored by TRACKING.KP_CONF_TYPE.
"""
import json
import os
import os.path as osp

from detectandtrack_amd.core.config import cfg

coco_src_keypoints = ['nose', 'left_eye', 'right_eye', 'left_ear', 'right_ear', 'left_shoulder', 'right_shoulder',
                      'left_elbow', 'right_elbow', 'left_wrist', 'right_wrist', 'left_hip', 'right_hip', 'left_knee',
                      'right_knee', 'left_ankle', 'right_ankle']
posetrack_src_keypoints = ['nose', 'head_bottom', 'head_top', 'left_ear', 'right_ear', 'left_shoulder', 'right_shoulder',
                           'left_elbow', 'right_elbow', 'left_wrist', 'right_wrist', 'left_hip', 'right_hip', 'left_knee',
                           'right_knee', 'left_ankle', 'right_ankle']
dst_keypoints = ['right_ankle', 'right_knee', 'right_hip', 'left_hip', 'left_knee', 'left_ankle', 'right_wrist', 'right_elbow',
                 'right_shoulder', 'left_shoulder', 'left_elbow', 'left_wrist', 'neck', 'nose', 'head_top']


def _compute_score(conf, global_conf):
    """:86-100"""
    t = cfg.TRACKING.KP_CONF_TYPE
    if t == 'global':
        return global_conf
    if t == 'local':
        return conf
    if t == 'scaled':
        return conf * global_conf
    raise NotImplementedError('Uknown type {}'.format(t))


def coco2posetrack(preds, src_kps, dst_kps, global_score):
    """:103-152.  preds: 4 x 17 rows (x, y, logit, prob) of one detection."""
    data = []
    global_score = float(global_score)
    thr = cfg.EVAL.EVAL_MPII_KPT_THRESHOLD

    def emit(k, x, y, local):
        if local >= thr:
            data.append({'id': [k], 'x': [float(x)], 'y': [float(y)], 'score': [float(_compute_score(local, global_score))]})
    for k, name in enumerate(dst_kps):
        if name in src_kps:
            i = src_kps.index(name)
            emit(k, preds[0, i], preds[1, i], (preds[2, i] + preds[2, i]) / 2.0)
        elif name in ('neck', 'head_top'):
            r, l = src_kps.index('right_shoulder'), src_kps.index('left_shoulder')
            xm, ym = (preds[0, r] + preds[0, l]) / 2.0, (preds[1, r] + preds[1, l]) / 2.0
            local = (preds[2, r] + preds[2, l]) / 2.0
            if name == 'neck':
                emit(k, xm, ym, local)
            else:
                n = src_kps.index('nose')
                emit(k, preds[0, n] - (xm - preds[0, n]), preds[1, n] - (ym - preds[1, n]), local)
    return data


def convert_data_to_annorect_struct(boxes, poses, tracks):
    """:155-187: one frame's detections (n x 5 boxes, n poses 4 x 17, n track ids) -> poseval 'annorect' list."""
    out = []
    for j in range(boxes.shape[0]):
        score = boxes[j, -1]
        if score < cfg.EVAL.EVAL_MPII_DROP_DETECTION_THRESHOLD:
            continue
        out.append({'annopoints': [{'point': coco2posetrack(poses[j], posetrack_src_keypoints, dst_keypoints, score)}],
                    'score': [float(score)], 'track_id': [tracks[j]]})
    if boxes.shape[0] == 0:   # MOTA needs at least one detection per image: the reference's dummy prediction
        out.append({'annopoints': [{'point': [{'id': [0], 'x': [0], 'y': [0], 'score': [-100.0]}]}], 'score': [0],
                    'track_id': [0]})
    return out


def write_posetrack_json(image_names, dets, output_dir, out_filenames=None):
    """:236-285 without the dataset lookups: `image_names[i]` = 'images/<video>/<frame>.jpg' of detection index i,
    `dets` = the detections(.withTracks) dict.  One JSON per video; `out_filenames` maps 'images/<video>' to the file name
    poseval expects (reference: derived from the annotation directory), default '<video with / -> _>.json'."""
    os.makedirs(output_dir, exist_ok=True)
    has_tracks = 'all_tracks' in dets
    per_video = {}
    for i, image_name in enumerate(image_names):
        video = osp.dirname(image_name)
        frame_num = int(osp.basename(image_name).split('.')[0])
        boxes, kps = dets['all_boxes'][1][i], dets['all_keyps'][1][i]
        tracks = dets['all_tracks'][1][i] if has_tracks else [1] * len(kps)
        per_video.setdefault(video, []).append({'image': image_name, 'imagenum': [frame_num],
                                                'annorect': convert_data_to_annorect_struct(boxes, kps, tracks)})
    written = []
    for video, vdata in per_video.items():
        name = (out_filenames or {}).get(video, video.replace('images/', '', 1).replace('/', '_') + '.json')
        path = osp.join(output_dir, name)
        with open(path, 'w') as f:
            json.dump({'annolist': vdata}, f)
        written.append(path)
    return written
