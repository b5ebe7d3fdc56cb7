"""COCO-format dataset -> roidb: the plumbing either side of the hot path that a real-data run needs.

Mirror of reference lib/datasets/json_dataset.py: the dataset catalogue (:35-63), `JsonDataset` (:70-108), `get_roidb` (:110-145), the
entry layout of `_prep_roidb_entry` (:147-174), ground-truth boxes / tracks / head boxes / keypoints of `_add_gt_annotations` (:203-301),
proposals from a file (:303-331, :424-476), the crowd filter (:479-497) and the class assignments (:500-516).  The reference reads the
annotation file through pycocotools (absent here, and only used as an index over the JSON): `CocoIndex` below is that index -- images,
annotations and categories by id, in the API's own iteration orders.  Weakly-labelled video shards (`_split_roidb_frames`, :378-421: files
that are not part of the release) are not restated.

A roidb entry made here is what `utils.video.get_clip` (clip ground truth), `roi_data.*` (training labels), `core.test_engine`
(inference: `image` paths decoded by read_frame) and `core.tracking_engine` (video / frame names) consume.
"""
import copy
import json
import logging
import os
import pickle

import numpy as np
import scipy.sparse

import detectandtrack_amd.utils.boxes as box_utils
from detectandtrack_amd.core.config import cfg

logger = logging.getLogger(__name__)

IM_DIR = 'image_directory'
ANN_FN = 'annotation_file'
ANN_DN = 'annotation_directory'
FRAMES_FROM_VIDEO = 'frames_from_video'

# the catalogue of the release (json_dataset.py:47-63), paths relative to the working directory like the reference's
DATASETS = {
    'posetrack_v1.0_train': {IM_DIR: 'lib/datasets/data/PoseTrack/', ANN_FN: 'lib/datasets/lists/PoseTrack/v1.0/posetrack_train.json',
                             ANN_DN: 'lib/datasets/data/PoseTrackV1.0_Annots_train_json/'},
    'posetrack_v1.0_val': {IM_DIR: 'lib/datasets/data/PoseTrack/', ANN_FN: 'lib/datasets/lists/PoseTrack/v1.0/posetrack_val.json',
                           ANN_DN: 'lib/datasets/data/PoseTrackV1.0_Annots_val_json'},
    'posetrack_v1.0_test': {IM_DIR: 'lib/datasets/data/PoseTrack/', ANN_FN: 'lib/datasets/lists/PoseTrack/v1.0/posetrack_test.json',
                            ANN_DN: 'lib/datasets/data/PoseTrackV1.0_Annots_test_json'},
}


def register(name, image_directory, annotation_file, annotation_directory=''):
    """Add / redirect a catalogue entry (a dataset mounted somewhere else than the reference's relative paths; the synthetic
    COCO-format directories of the tests).  DAT_DATASET_ROOT=<dir> in the environment prefixes the release's relative paths instead."""
    DATASETS[name] = {IM_DIR: image_directory, ANN_FN: annotation_file, ANN_DN: annotation_directory}


def load_catalog_from_env():
    """DAT_DATASET_CATALOG=<json file> {name: {image_directory, annotation_file[, annotation_directory]}}: catalogue entries for datasets
    mounted outside the reference's relative paths (tools/test_net.py, tools/compute_tracks.py and tools/train_net.py call this)."""
    path = os.environ.get('DAT_DATASET_CATALOG', '')
    if path:
        with open(path) as f:
            for name, d in json.load(f).items():
                register(name, d[IM_DIR], d[ANN_FN], d.get(ANN_DN, ''))


def _resolve(path):
    root = os.environ.get('DAT_DATASET_ROOT', '')
    return os.path.join(root, path) if (root and not os.path.isabs(path)) else path


class CocoIndex(object):
    """What pycocotools.coco.COCO is to json_dataset.py: an index over one annotation file.  getImgIds / getCatIds return the ids in file
    order (the API keeps dict insertion order; the reference sorts the image ids itself, :116-117), getAnnIds(imgIds=i) the image's
    annotations in file order (`imgToAnns`), iscrowd=None meaning no filter."""

    def __init__(self, annotation_file):
        with open(annotation_file) as f:
            self.dataset = json.load(f)
        self.imgs = {im['id']: im for im in self.dataset.get('images', [])}
        self.cats = {c['id']: c for c in self.dataset.get('categories', [])}
        self.anns, self.imgToAnns = {}, {}
        for a in self.dataset.get('annotations', []):
            self.anns[a['id']] = a
            self.imgToAnns.setdefault(a['image_id'], []).append(a)

    def getImgIds(self):
        return list(self.imgs.keys())

    def loadImgs(self, ids):
        return [self.imgs[i] for i in ids]

    def getCatIds(self):
        return list(self.cats.keys())

    def loadCats(self, ids):
        return [self.cats[i] for i in ids]

    def getAnnIds(self, imgIds, iscrowd=None):
        ids = imgIds if isinstance(imgIds, (list, tuple)) else [imgIds]
        anns = [a for i in ids for a in self.imgToAnns.get(i, [])]
        if iscrowd is not None:
            anns = [a for a in anns if a['iscrowd'] == iscrowd]
        return [a['id'] for a in anns]

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids]


class JsonDataset(object):
    """:70-108."""

    def __init__(self, name):
        assert name in DATASETS.keys(), 'Unknown dataset name'
        self.name = name
        self.image_directory = _resolve(DATASETS[name][IM_DIR])
        self.COCO = CocoIndex(_resolve(DATASETS[name][ANN_FN]))
        self.annotation_directory = _resolve(DATASETS[name][ANN_DN]) if DATASETS[name].get(ANN_DN) else ''
        category_ids = self.COCO.getCatIds()
        categories = [c['name'] for c in self.COCO.loadCats(category_ids)]
        self.category_to_id_map = dict(zip(categories, category_ids))
        self.classes = ['__background__'] + categories
        self.num_classes = len(self.classes)
        self.json_category_id_to_contiguous_id = {v: i + 1 for i, v in enumerate(self.COCO.getCatIds())}
        self.contiguous_category_id_to_json_id = {v: k for k, v in self.json_category_id_to_contiguous_id.items()}
        self._init_keypoints(name=self.name)
        self.person_cat_info = self.COCO.loadCats([self.category_to_id_map['person']])[0]     # (head keypoints for the PCK distance)
        self.frames_from_video = bool(DATASETS[name].get(FRAMES_FROM_VIDEO, False))
        self.annotations_info = None

    def get_roidb(self, gt=False, proposal_file=None, min_proposal_size=2, proposal_limit=-1, crowd_filter_thresh=0):
        """:110-145."""
        assert gt is True or crowd_filter_thresh == 0, \
            'Crowd filter threshold must be 0 if ground-truth annotations are not included.'
        image_ids = self.COCO.getImgIds()
        image_ids.sort()
        roidb = copy.deepcopy(self.COCO.loadImgs(image_ids))
        if len(cfg.ROIDB_SUBSET) > 0:
            roidb = roidb[cfg.ROIDB_SUBSET[0]: cfg.ROIDB_SUBSET[1]]
            logger.warning('Using a roidb subset {}'.format(cfg.ROIDB_SUBSET))
        for entry in roidb:
            self._prep_roidb_entry(entry)
        if gt:
            for entry in roidb:
                self._add_gt_annotations(entry)
        if proposal_file is not None:
            self._add_proposals_from_file(roidb, proposal_file, min_proposal_size, proposal_limit, crowd_filter_thresh)
        _add_class_assignments(roidb)
        return roidb

    def _row_fields(self):
        """(name, trailing shape, dtype) of the per-box arrays of an entry (:153-168)"""
        f = [('boxes', (4,), np.float32), ('tracks', (1,), np.int32), ('head_boxes', (4,), np.float32), ('gt_classes', (), np.int32),
             ('seg_areas', (), np.float32), ('is_crowd', (), bool), ('box_to_gt_ind_map', (), np.int32)]
        if self.keypoints is not None:
            f.append(('gt_keypoints', (3, self.num_keypoints), np.int32))
        return f

    def _prep_roidb_entry(self, entry):
        """:147-174: the image record of the JSON becomes the entry; empty per-box arrays; `image` = path of the file."""
        entry.update(dataset=self, image=os.path.join(self.image_directory, entry['file_name']), flipped=False,
                     has_visible_keypoints=False, segms=[])
        for name, tail, dt in self._row_fields():
            entry[name] = np.empty((0,) + tail, dtype=dt)
        entry['gt_overlaps'] = scipy.sparse.csr_matrix(np.empty((0, self.num_classes), dtype=np.float32))
        for k in ('date_captured', 'url', 'license', 'file_name'):
            entry.pop(k, None)

    @staticmethod
    def _clean_box(obj, height, width):
        """x, y, w, h of the JSON -> x1, y1, x2, y2 clipped to the image, or None for an annotation the reference drops (:214-233):
        area below TRAIN.GT_MIN_AREA, `ignore`, zero area, a box that is not wider AND taller than one pixel after clipping."""
        if obj['area'] < cfg.TRAIN.GT_MIN_AREA or obj.get('ignore', 0) == 1:
            return None
        x, y, w, h = obj['bbox'][:4]
        x1, y1, x2, y2 = clip_xyxy_to_image(x, y, x + np.maximum(0., w - 1.), y + np.maximum(0., h - 1.), height, width)
        return [x1, y1, x2, y2] if (obj['area'] > 0 and x2 > x1 and y2 > y1) else None

    def _add_gt_annotations(self, entry):
        """:203-301 on the annotations of the JSON (the weak-annotation branch belongs to _split_roidb_frames): one row per usable
        annotation, in file order."""
        objs = self.COCO.loadAnns(self.COCO.getAnnIds(imgIds=entry['id'], iscrowd=None))
        rows = []
        for obj in objs:
            if isinstance(obj['segmentation'], list):           # polygons with fewer than 3 points go; crowd regions are RLE dicts
                obj['segmentation'] = [p for p in obj['segmentation'] if len(p) >= 6]
            box = self._clean_box(obj, entry['height'], entry['width'])
            if box is not None:
                obj['clean_bbox'] = box
                rows.append(obj)
        n = len(rows)
        cls = np.array([self.json_category_id_to_contiguous_id[o['category_id']] for o in rows], dtype=np.int32).reshape(n)
        crowd = np.array([bool(o['iscrowd']) for o in rows], dtype=bool).reshape(n)
        cols = {'boxes': np.array([o['clean_bbox'] for o in rows], dtype=np.float32).reshape(n, 4),
                'tracks': np.array([o.get('track_id', -1) for o in rows], dtype=np.int32).reshape(n, 1),
                # (head boxes stay as the JSON has them -- not cleaned, not converted: only the MPII evaluation reads them)
                'head_boxes': np.array([o.get('head_box', [-1] * 4) for o in rows], dtype=np.float32).reshape(n, 4),
                'gt_classes': cls, 'seg_areas': np.array([o['area'] for o in rows], dtype=np.float32).reshape(n),
                'is_crowd': crowd, 'box_to_gt_ind_map': np.arange(n, dtype=np.int32)}
        overlaps = np.zeros((n, self.num_classes), dtype=np.float32)
        overlaps[np.arange(n), cls] = 1.0
        overlaps[crowd, :] = -1.0                               # crowd regions: excluded from training through every class
        if self.keypoints is not None:
            cols['gt_keypoints'] = np.array([self._get_gt_keypoints(o) for o in rows], dtype=np.int32).reshape(n, 3, self.num_keypoints)
            entry['has_visible_keypoints'] = bool(n and cols['gt_keypoints'][:, 2, :].sum(axis=1).max() > 0)
        _append_rows(entry, cols, overlaps)
        entry['segms'].extend(o['segmentation'] for o in rows)

    def _add_proposals_from_file(self, roidb, proposal_file, min_proposal_size, top_k, crowd_thresh):
        """:303-331."""
        with open(proposal_file, 'rb') as f:
            proposals = pickle.load(f, encoding='latin1')
        id_field = 'indexes' if 'indexes' in proposals else 'ids'
        _sort_proposals(proposals, id_field)
        box_list = []
        for i, entry in enumerate(roidb):
            boxes = proposals['boxes'][i]
            assert entry['id'] == proposals[id_field][i]
            boxes = clip_boxes_to_image(boxes, entry['height'], entry['width'])
            boxes = boxes[unique_boxes(boxes), :]
            boxes = boxes[filter_small_boxes(boxes, min_proposal_size), :]
            if top_k > 0:
                boxes = boxes[:top_k, :]
            box_list.append(boxes)
        _merge_proposal_boxes_into_roidb(roidb, box_list)
        if crowd_thresh > 0:
            _filter_crowd_proposals(roidb, crowd_thresh)

    def _init_keypoints(self, name=''):
        """:333-366."""
        self.keypoints = None
        self.keypoint_flip_map = None
        self.keypoints_to_id_map = None
        self.num_keypoints = 0
        if 'person' not in self.category_to_id_map:
            return
        cat_info = self.COCO.loadCats([self.category_to_id_map['person']])
        if 'keypoints' in cat_info[0]:
            keypoints = cat_info[0]['keypoints']
            self.keypoints_to_id_map = dict(zip(keypoints, range(len(keypoints))))
            self.keypoints = keypoints
            self.num_keypoints = len(keypoints)
            sides = ['shoulder', 'elbow', 'wrist', 'hip', 'knee', 'ankle']
            if name.startswith('keypoints_coco'):
                sides = ['eye', 'ear'] + sides
            self.keypoint_flip_map = {'left_' + s: 'right_' + s for s in sides}

    def _get_gt_keypoints(self, obj):
        """:368-385: (3, K) int32 rows x, y, visibility (0 not labelled, 1 labelled outside the mask, 2 labelled inside)."""
        if 'keypoints' not in obj:
            return None
        kp = np.array(obj['keypoints'])
        assert len(obj['keypoints']) / 3 == self.num_keypoints
        gt_kps = np.ones((3, self.num_keypoints), dtype=np.int32)
        gt_kps[0], gt_kps[1], gt_kps[2] = kp[0::3], kp[1::3], kp[2::3]
        return gt_kps


# ---- box helpers of lib/utils/boxes.py used only here (:81-87, :118-138) -------------------------------------------------------------
def unique_boxes(boxes, scale=1.0):
    v = np.array([1, 1e3, 1e6, 1e9])
    hashes = np.round(boxes * scale).dot(v)
    _, index = np.unique(hashes, return_index=True)
    return np.sort(index)


def filter_small_boxes(boxes, min_size):
    w = boxes[:, 2] - boxes[:, 0]
    h = boxes[:, 3] - boxes[:, 1]
    return np.where((w >= min_size) & (h > min_size))[0]        # (the asymmetry is the reference's)


def clip_boxes_to_image(boxes, height, width):
    boxes[:, [0, 2]] = np.minimum(width - 1., np.maximum(0., boxes[:, [0, 2]]))
    boxes[:, [1, 3]] = np.minimum(height - 1., np.maximum(0., boxes[:, [1, 3]]))
    return boxes


def clip_xyxy_to_image(x1, y1, x2, y2, height, width):
    x1 = np.minimum(width - 1., np.maximum(0., x1))
    y1 = np.minimum(height - 1., np.maximum(0., y1))
    x2 = np.minimum(width - 1., np.maximum(0., x2))
    y2 = np.minimum(height - 1., np.maximum(0., y2))
    return x1, y1, x2, y2


def _append_rows(entry, cols, overlaps):
    """Append rows to every per-box array an entry has (arrays the caller does not give are extended by zeros) and to gt_overlaps."""
    n = overlaps.shape[0]
    for name, cur in list(entry.items()):
        if name in cols:
            entry[name] = np.append(cur, cols[name].astype(cur.dtype, copy=False), axis=0)
        elif name in ('gt_classes', 'seg_areas', 'is_crowd'):
            entry[name] = np.append(cur, np.zeros((n,) + cur.shape[1:], dtype=cur.dtype), axis=0)
    entry['gt_overlaps'] = scipy.sparse.csr_matrix(np.append(entry['gt_overlaps'].toarray(), overlaps.astype(np.float32), axis=0))


def _merge_proposal_boxes_into_roidb(roidb, box_list):
    """:424-476: proposal rows behind the ground truth -- class 0, overlap = the best IoU with a gt box (crowd gt included on purpose:
    _filter_crowd_proposals deals with those) recorded under that box's class, box_to_gt_ind_map = that box (-1: no overlap)."""
    assert len(box_list) == len(roidb)
    for entry, boxes in zip(roidb, box_list):
        m = boxes.shape[0]
        overlaps = np.zeros((m, entry['gt_overlaps'].shape[1]), dtype=np.float32)
        to_gt = -np.ones((m,), dtype=np.int32)
        gt_inds = np.where(entry['gt_classes'] > 0)[0]
        if len(gt_inds) > 0:
            iou = box_utils.bbox_overlaps(boxes.astype(np.float32, copy=False), entry['boxes'][gt_inds, :].astype(np.float32, copy=False))
            best, val = iou.argmax(axis=1), iou.max(axis=1)
            hit = np.where(val > 0)[0]
            overlaps[hit, entry['gt_classes'][gt_inds][best[hit]]] = val[hit]
            to_gt[hit] = gt_inds[best[hit]]
        _append_rows(entry, {'boxes': boxes, 'box_to_gt_ind_map': to_gt}, overlaps)


def crowd_iou(dt_xywh, gt_xywh):
    """pycocotools.mask.iou(dt, gt, iscrowd = all True) on [x, y, w, h] boxes (maskApi.c bbIou): intersection over the area of the
    DETECTION (a crowd region absorbs what lies inside it), widths / heights as they are (no + 1)."""
    dt, gt = np.asarray(dt_xywh, dtype=np.float64), np.asarray(gt_xywh, dtype=np.float64)
    out = np.zeros((dt.shape[0], gt.shape[0]), dtype=np.float64)
    for j in range(gt.shape[0]):
        w = np.minimum(dt[:, 0] + dt[:, 2], gt[j, 0] + gt[j, 2]) - np.maximum(dt[:, 0], gt[j, 0])
        h = np.minimum(dt[:, 1] + dt[:, 3], gt[j, 1] + gt[j, 3]) - np.maximum(dt[:, 1], gt[j, 1])
        inter = np.where((w > 0) & (h > 0), w * h, 0.0)
        out[:, j] = inter / (dt[:, 2] * dt[:, 3])
    return out


def _filter_crowd_proposals(roidb, crowd_thresh):
    """:479-497: proposals inside crowd regions get overlap -1 with every class (excluded from training)."""
    for entry in roidb:
        gt_overlaps = entry['gt_overlaps'].toarray()
        crowd_inds = np.where(entry['is_crowd'] == 1)[0]
        non_gt_inds = np.where(entry['gt_classes'] == 0)[0]
        if len(crowd_inds) == 0 or len(non_gt_inds) == 0:
            continue
        crowd_boxes = box_utils.xyxy_to_xywh(entry['boxes'][crowd_inds, :])
        non_gt_boxes = box_utils.xyxy_to_xywh(entry['boxes'][non_gt_inds, :])
        ious = crowd_iou(non_gt_boxes, crowd_boxes)
        bad_inds = np.where(ious.max(axis=1) > crowd_thresh)[0]
        gt_overlaps[non_gt_inds[bad_inds], :] = -1
        entry['gt_overlaps'] = scipy.sparse.csr_matrix(gt_overlaps)


def _add_class_assignments(roidb):
    """:500-516."""
    for entry in roidb:
        gt_overlaps = entry['gt_overlaps'].toarray()
        max_overlaps = gt_overlaps.max(axis=1)
        max_classes = gt_overlaps.argmax(axis=1)
        entry['max_classes'] = max_classes
        entry['max_overlaps'] = max_overlaps
        assert all(max_classes[np.where(max_overlaps == 0)[0]] == 0)        # no overlap: background
        assert all(max_classes[np.where(max_overlaps > 0)[0]] != 0)         # overlap: a foreground class


def _sort_proposals(proposals, id_field):
    order = np.argsort(proposals[id_field])
    for k in ['boxes', id_field, 'scores']:
        proposals[k] = [proposals[k][i] for i in order]
