"""A real-data run made possible: `tools/test_net.py --cfg <a shipped reference yaml>` resolves TEST.DATASET through the
dataset layer (datasets/json_dataset.py -> utils/video.get_clip), decodes frame FILES, runs the pipelined engine (mixed resolutions, per-frame
trunk cache), `tools/compute_tracks.py` reads the same dataset with its ground truth and writes detections_withTracks.pkl -- end to end
on a synthetic COCO-format dataset DIRECTORY (annotation JSON + image files), and equal to the reference-shaped eager loop
(cfg.HIP.PIPELINE_DEPTH 0: one clip at a time through im_detect_all) -- bit for bit with one clip per forward (the same kernels on the
same grids; several clips per forward change the summation order of the conv grids, tests/test_gpu_model.py)."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _nested(flat):
    out = {}
    for k, v in flat.items():
        d, parts = out, k.split('.')
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    return out


def _make_dataset(root, seed=2):
    """images/<video>/<frame>.png + a PoseTrack-shaped annotation file: two videos of different resolution, 5 and 4 frames."""
    from PIL import Image
    rs = np.random.RandomState(seed)
    images, anns, aid = [], [], 1
    names = ['nose', 'head_bottom', 'head_top', 'left_ear', 'right_ear', 'left_shoulder', 'right_shoulder', 'left_elbow', 'right_elbow',
             'left_wrist', 'right_wrist', 'left_hip', 'right_hip', 'left_knee', 'right_knee', 'left_ankle', 'right_ankle']
    for vi, (vid, n, (h, w)) in enumerate((('000001_bonn', 5, (180, 320)), ('000002_mpii', 4, (200, 260)))):
        os.makedirs(os.path.join(root, 'images', vid))
        base = rs.randint(0, 255, (h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
        for f in range(1, n + 1):
            img = np.repeat(np.repeat(base, 8, 0), 8, 1)[:h, :w].astype(np.int32) + rs.randint(-20, 20, (h, w, 3)) + 3 * f
            Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(root, 'images', vid, '%06d.png' % f))
            iid = 100 * (vi + 1) + f
            images.append({'id': iid, 'file_name': 'images/%s/%06d.png' % (vid, f), 'width': w, 'height': h, 'nframes': n, 'frame_id': f,
                           'is_labeled': True})
            for tid in range(2):
                bx = [20. + 60 * tid + 2 * f, 15. + 5 * tid, 70., 120.]
                kp = []
                for k in range(17):
                    kp += [int(bx[0] + rs.uniform(0, 1) * bx[2]), int(bx[1] + rs.uniform(0, 1) * bx[3]), 2]
                anns.append({'id': aid, 'image_id': iid, 'category_id': 1, 'bbox': bx, 'area': bx[2] * bx[3], 'iscrowd': 0, 'keypoints': kp,
                             'num_keypoints': 17, 'track_id': tid, 'segmentation': [], 'head_box': [bx[0], bx[1], bx[0] + 20, bx[1] + 20]})
                aid += 1
    ann = os.path.join(root, 'posetrack_val.json')
    with open(ann, 'w') as f:
        json.dump({'images': images, 'annotations': anns, 'categories': [{'id': 1, 'name': 'person', 'keypoints': names}]}, f)
    cat = os.path.join(root, 'catalog.json')
    with open(cat, 'w') as f:
        json.dump({'posetrack_v1.0_val': {'image_directory': root, 'annotation_file': ann, 'annotation_directory': os.path.join(root, 'annots')}}, f)
    return cat, len(images)


@pytest.mark.parametrize('rel', ['video/3d/04_R-18-3D_PTFromImNet.yaml', 'video/2d_best/01_R101_best_hungarian.yaml'])
def test_tools_run_a_shipped_config_on_a_dataset_directory(tmp_path, rel):
    with open(os.path.join(REPO, 'tests', 'golden', 'reference_cfg_files.json')) as f:
        shipped = json.load(f)[rel]
    # the shipped file as the real lib/core/config.py reads it (the GPU box has no /root/reference): written back as a yaml
    flat = {k: v for k, v in shipped.items() if k not in ('RPN.ON', 'VIDEO.NUM_FRAMES_MID')}        # (derived by assert_and_infer_cfg)
    cfg_file = str(tmp_path / 'shipped.yaml')
    with open(cfg_file, 'w') as f:
        yaml.safe_dump(_nested(flat), f)
    catalog, n_images = _make_dataset(str(tmp_path / 'data'))
    env = dict(os.environ, DAT_DATASET_CATALOG=catalog, PYTHONPATH=REPO)

    def run(tag, extra):
        out = str(tmp_path / tag)
        os.makedirs(out)
        # overrides a run on random weights needs: no checkpoint, detections kept for the tracker, the box's one GPU
        opts = ['OUTPUT_DIR', out, 'NUM_GPUS', '1', 'TEST.WEIGHTS', '', 'TRACKING.CONF_FILTER_INITIAL_DETS', '0.0', 'TEST.SCORE_THRESH', '0.0',
                'TEST.DETECTIONS_PER_IM', '12', 'HIP.DTYPE', 'fp32', 'RNG_SEED', '3'] + extra
        for tool, more in (('test_net.py', ['--synthetic-weights']), ('compute_tracks.py', [])):
            p = subprocess.run([sys.executable, os.path.join(REPO, 'tools', tool), '--cfg', cfg_file] + more + opts, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            assert p.returncode == 0, (tool, p.stderr.decode()[-3000:])
            if tool == 'test_net.py':
                stats = [json.loads(ln)['test_net'] for ln in p.stdout.decode().splitlines() if ln.startswith('{"test_net"')]
        sub = os.path.join(out, 'test', 'posetrack_v1.0_val', 'keypoint_rcnn')
        with open(os.path.join(sub, 'detections_withTracks.pkl'), 'rb') as f:
            return pickle.load(f), (stats[0] if stats else None)
    three_d = '-3D_' in rel
    piped, st = run('pipelined', ['HIP.PIPELINE_DEPTH', '3', 'HIP.IMS_PER_FORWARD', '1'] + (['HIP.FRAME_TRUNK_CACHE', '8'] if three_d else []))
    eager, st0 = run('eager', ['HIP.PIPELINE_DEPTH', '0'])
    assert st is not None and st['clips'] == n_images and st0 is None
    if three_d:
        assert st['frame_trunk_cache'] > 0 and st['trunk_resets'] == 1 and st['trunk_frames_computed'] == n_images     # every FILE decoded / run once
    assert sorted(piped) == sorted(eager) and 'all_tracks' in piped
    n_det = 0
    for i in range(n_images):
        a, b = piped['all_boxes'][1][i], eager['all_boxes'][1][i]
        np.testing.assert_array_equal(a, b, err_msg='clip %d' % i)
        assert len(piped['all_keyps'][1][i]) == len(eager['all_keyps'][1][i])
        for x, y in zip(piped['all_keyps'][1][i], eager['all_keyps'][1][i]):
            np.testing.assert_array_equal(x, y)
        assert list(piped['all_tracks'][1][i]) == list(eager['all_tracks'][1][i])
        n_det += len(a)
    assert n_det > 0
