"""The dataset layer (detectandtrack_amd/datasets/json_dataset.py, utils/video.get_clip / combine_clip) against the REFERENCE's own
lib/datasets/json_dataset.py and lib/utils/video.py run on tests/golden/synthetic_posetrack.json (tests/golden/make_golden.py
golden_dataset, through a stub of the four pycocotools calls): every field of every roidb entry -- with and without ground truth, with
proposals from a file -- and of every clip entry (tube boxes, keypoints as t * K + k, track visibility) must be identical."""
import json
import os

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, 'tests', 'golden')
ARRAYS = ('boxes', 'tracks', 'head_boxes', 'gt_classes', 'seg_areas', 'is_crowd', 'box_to_gt_ind_map', 'gt_keypoints', 'max_classes',
          'max_overlaps', 'track_visible')
SCALARS = ('id', 'width', 'height', 'nframes', 'frame_id', 'is_labeled', 'flipped', 'has_visible_keypoints')


@pytest.fixture()
def dataset():
    from detectandtrack_amd.core.config import reset_cfg
    from detectandtrack_amd.datasets import json_dataset as jd
    reset_cfg()
    jd.register('synthetic_posetrack', '/data/PoseTrack/', os.path.join(GOLD, 'synthetic_posetrack.json'), '/data/annots')
    yield jd.JsonDataset('synthetic_posetrack')
    jd.DATASETS.pop('synthetic_posetrack', None)
    reset_cfg()


def _check(roidb, arrays, meta, prefix):
    assert len(roidb) == len(meta), (prefix, len(roidb), len(meta))
    for i, (e, m) in enumerate(zip(roidb, meta)):
        assert sorted(str(k) for k in e.keys()) == m['keys'], (prefix, i, sorted(set(map(str, e.keys())) ^ set(m['keys'])))
        for k in ARRAYS:
            key = '%s/%d/%s' % (prefix, i, k)
            assert (k in e) == (key in arrays), key
            if k in e:
                ref = arrays[key]
                got = np.asarray(e[k])
                assert got.dtype == ref.dtype and got.shape == ref.shape, (key, got.dtype, ref.dtype, got.shape, ref.shape)
                np.testing.assert_array_equal(got, ref, err_msg=key)
        ov = e['gt_overlaps'].toarray()
        ref = arrays['%s/%d/gt_overlaps' % (prefix, i)]
        assert ov.dtype == ref.dtype
        np.testing.assert_array_equal(ov, ref)
        for k in SCALARS:
            assert (k in e and not isinstance(e[k], np.ndarray)) == (k in m), (prefix, i, k)
            if k in m:
                assert (bool(e[k]) if isinstance(m[k], bool) else int(e[k])) == m[k], (prefix, i, k)
        assert e['image'] == m['image'] and [len(s) for s in e['segms']] == m['n_segms']
        for k in ('all_frame_ids', 'original_file_name'):
            assert (k in e) == (k in m)
            if k in m:
                assert (list(e[k]) if isinstance(m[k], list) else e[k]) == m[k]


def test_json_dataset_roidb_matches_the_reference(dataset):
    arrays = np.load(os.path.join(GOLD, 'reference_json_dataset.npz'))
    with open(os.path.join(GOLD, 'reference_json_dataset.json')) as f:
        meta = json.load(f)
    d = meta['dataset']
    assert dataset.classes == d['classes'] and dataset.num_classes == d['num_classes'] and dataset.keypoints == d['keypoints']
    assert dataset.num_keypoints == d['num_keypoints'] == 17 and dataset.keypoint_flip_map == d['keypoint_flip_map']
    assert dataset.category_to_id_map == d['category_to_id_map'] and sorted(dataset.person_cat_info) == d['person_cat_info_keys']
    assert dataset.image_directory == d['image_directory'] and dataset.annotation_directory == d['annotation_directory']
    assert dataset.frames_from_video == d['frames_from_video']
    gt = dataset.get_roidb(gt=True)
    _check(gt, arrays, meta['gt'], 'gt')
    _check(dataset.get_roidb(gt=False), arrays, meta['nogt'], 'nogt')
    # what the fixture is for: ids sorted, the sanitiser's drops, the crowd row, the empty image
    assert [e['id'] for e in gt] == sorted(e['id'] for e in gt)
    assert any(len(e['boxes']) == 0 for e in gt) and any(e['is_crowd'].any() for e in gt)
    crowd = next(e for e in gt if e['is_crowd'].any())
    assert (crowd['gt_overlaps'].toarray()[crowd['is_crowd']] == -1).all() and (crowd['max_overlaps'][crowd['is_crowd']] == -1).all()
    _check(dataset.get_roidb(gt=True, proposal_file=os.path.join(GOLD, 'synthetic_posetrack_proposals.pkl'), min_proposal_size=2,
                             proposal_limit=8), arrays, meta['props'], 'props')


def test_get_clip_tube_ground_truth_matches_the_reference(dataset):
    from detectandtrack_amd.core.config import cfg
    from detectandtrack_amd.utils import video as video_utils
    arrays = np.load(os.path.join(GOLD, 'reference_json_dataset.npz'))
    with open(os.path.join(GOLD, 'reference_json_dataset.json')) as f:
        meta = json.load(f)
    assert len(meta['clips']) == 6
    for ci, case in enumerate(meta['clips']):
        cfg.VIDEO.NUM_FRAMES, cfg.VIDEO.NUM_FRAMES_MID, cfg.VIDEO.TIME_INTERVAL = case['T'], case['mid'], case['time_interval']
        clips = video_utils.get_clip(dataset.get_roidb(gt=True), remove_imperfect=case['remove_imperfect'])
        for c, m in zip(clips, case['entries']):
            assert c['image'] == m['image'] and len(c['image']) == case['T']
        _check(clips, arrays, case['entries'], 'clips%d' % ci)
        if case['T'] == 3 and case['mid'] == 3 and not case['remove_imperfect'] and case['time_interval'] == 1:
            # the frame list of get_clip agrees with the frame-number rule the synthetic-video tools use (clip_frame_ids) wherever every
            # frame of the video exists
            by_video = {}
            for c in clips:
                by_video.setdefault(os.path.dirname(c['image'][1]), []).append(c)
            full = by_video['/data/PoseTrack/images/bonn_000001']
            for c in full:
                key = int(os.path.basename(c['image'][1])[:-4])
                assert [int(os.path.basename(p)[:-4]) for p in c['image']] == video_utils.clip_frame_ids(key, 1, 5, 3)


def test_crowd_iou_known_answers():
    """datasets.json_dataset.crowd_iou = pycocotools.mask.iou with iscrowd set (maskApi.c bbIou: intersection / area of the DETECTION; the
    one piece of the dataset layer whose reference code is not in the tree): hand-worked cases."""
    from detectandtrack_amd.datasets.json_dataset import crowd_iou
    dt = np.array([[0, 0, 10, 10], [5, 5, 10, 10], [20, 20, 4, 4], [0, 0, 10, 20]], dtype=np.float64)
    gt = np.array([[0, 0, 10, 10], [8, 8, 100, 100]], dtype=np.float64)
    got = crowd_iou(dt, gt)
    want = np.array([[1.0, 4.0 / 100.0], [25.0 / 100.0, 49.0 / 100.0], [0.0, 1.0], [0.5, 2 * 12 / 200.0]])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)
