"""GenerateProposalLabels on the device (round 5): the roi sampling of a training iteration as ONE kernel
(`dat_sample_rois`) on the device-resident proposals of the clip, instead of a device -> host copy of 2000 proposals, ~2 ms of NumPy
(lib/roi_data/fast_rcnn.py:109-203 restated in roi_data/fast_rcnn.py) and a dozen host -> device uploads of the sampled blobs.

`DeviceRoiSampler(entry)` keeps the clip's ground truth (boxes, classes, keypoints) on the GPU; the training executor calls it from
CollectAndDistributeFpnRpnProposals / GenerateProposalLabels with the collected proposals still on the device.  What comes back to the
host is two integers -- how many rois and keypoint rois were drawn -- because the head launches that follow are sized by them (the
reference's Python op resizes its output blobs the same way).

Contract of the draw: see include/dat_hip.h `dat_sample_rois` (NumPy's Mersenne-Twister stream of `npr.choice` is replaced by a
counter-based key per (seed, iteration, candidate); same candidate sets, same counts, uniform without replacement)."""
import numpy as np
import torch

from detectandtrack_amd.core.config import cfg
from detectandtrack_amd.ops import hip_ops as ops


class DeviceRoiSampler(object):
    on_device = True
    MAX_CANDIDATES, MAX_T = 4096, 8         # RS_MAXN / LAB_MAXT of csrc/labels.hip

    def __init__(self, entry, seed=0, device=None):
        """seed: an int, or a zero-argument callable that is asked for one only when the device sampler is really built (the host sampler's
        numpy.random stream must not lose a draw to a sampler that is never used)."""
        gt = np.where(entry['gt_classes'] > 0)[0]
        # the device path covers what every clip of the training set looks like here: all rows of the entry are ground-truth boxes
        # (no crowd regions, no pre-computed proposals); anything else keeps the host sampler
        assert len(gt) == len(entry['gt_classes']) and not np.any(entry['is_crowd']), 'entry with non-gt rows: use the host sampler'
        assert np.array_equal(entry['box_to_gt_ind_map'], np.arange(len(gt))), 'gt rows must map to themselves'
        assert cfg.TRAIN.FG_THRESH >= cfg.TRAIN.BG_THRESH_HI, 'overlapping fg / bg ranges are not supported on the device'
        # the kernel's own limits (csrc/labels.hip dat_sample_rois: RS_MAXN candidates, LAB_MAXT frames per tube, every gt a keypoint roi), checked
        # HERE against the configuration so that make_sampler falls back to the host restatement instead of a launch error mid-iteration
        # (ADVICE r5): the proposal blob holds at most RPN_POST_NMS_TOP_N rows (collect, one image per GPU)
        G = len(gt)
        self.T = entry['boxes'].shape[1] // 4
        fg_per_im = int(np.round(cfg.TRAIN.FG_FRACTION * cfg.TRAIN.BATCH_SIZE_PER_IM))
        assert 1 <= self.T <= self.MAX_T, 'tubes of %d frames: the device sampler holds up to %d' % (self.T, self.MAX_T)
        assert G >= 1 and G + int(cfg.TRAIN.RPN_POST_NMS_TOP_N) <= self.MAX_CANDIDATES, \
            '%d gt boxes + %d proposals exceed the device sampler\'s %d candidates' % (G, cfg.TRAIN.RPN_POST_NMS_TOP_N, self.MAX_CANDIDATES)
        assert not cfg.MODEL.KEYPOINTS_ON or G <= fg_per_im, '%d persons for %d foreground rois per image' % (G, fg_per_im)
        dev = device or torch.device('cuda', torch.cuda.current_device())
        self.gt_boxes = torch.from_numpy(np.ascontiguousarray(entry['boxes'], dtype=np.float32)).to(dev)
        self.gt_classes = torch.from_numpy(np.ascontiguousarray(entry['gt_classes'], dtype=np.int32)).to(dev)
        self.gt_kps = None
        if cfg.MODEL.KEYPOINTS_ON:
            self.gt_kps = torch.from_numpy(np.ascontiguousarray(entry['gt_keypoints'], dtype=np.int32)).to(dev)
        self.seed, self.iter = int(seed() if callable(seed) else seed), 0
        self.last_counts = None

    def __call__(self, rois_dev, n_dev, im_info, want_picked=False):
        """rois_dev fp32 [cap, 4T+1] (network scale), n_dev int32 device count; im_info host [1, 3].  -> dict blob name -> device tensor
        holding exactly the drawn rows."""
        per_im = int(cfg.TRAIN.BATCH_SIZE_PER_IM)
        fg_per_im = int(np.round(cfg.TRAIN.FG_FRACTION * per_im))
        K = int(cfg.KRCNN.NUM_KEYPOINTS) if self.gt_kps is not None else 0
        out = ops.sample_rois(rois_dev, n_dev.view(-1)[:1].to(torch.int32), self.gt_boxes, self.gt_classes, self.gt_kps, self.T,
                              int(cfg.MODEL.NUM_CLASSES), bool(cfg.MODEL.CLS_AGNOSTIC_BBOX_REG), K, int(cfg.KRCNN.HEATMAP_SIZE), per_im, fg_per_im,
                              cfg.TRAIN.FG_THRESH, cfg.TRAIN.BG_THRESH_HI, cfg.TRAIN.BG_THRESH_LO, cfg.MODEL.BBOX_REG_WEIGHTS,
                              float(im_info[0, 2]), self.seed, self.iter, want_picked=want_picked)
        self.iter += 1
        counts = out.pop('counts').cpu().numpy()            # the ONE read-back (32 bytes): rows drawn -- they size the head launches
        self.last_counts = counts
        n, m = int(counts[0]), int(counts[2])
        blobs = {k: out[k][:n] for k in ('rois', 'labels_int32', 'bbox_targets', 'bbox_inside_weights', 'bbox_outside_weights')}
        if self.gt_kps is not None:
            blobs['keypoint_rois'] = out['keypoint_rois'][:m]
            blobs['keypoint_locations_int32'] = out['keypoint_locations_int32'][:m].reshape(-1, 1)
            blobs['keypoint_weights'] = out['keypoint_weights'][:m].reshape(-1, 1)
            blobs['keypoint_loss_normalizer'] = torch.ones(1, dtype=torch.float32, device=rois_dev.device)
            blobs['keypoint_weights_sum'] = float(counts[6])     # (what the keypoint loss normalises by: no read-back of the weights)
        if want_picked:
            blobs['picked'] = out['picked']
        return blobs


class WeightSum(object):
    """Host-side stand-in of a device weight blob for consumers that only need its sum (training.op_KeypointLoss)."""

    def __init__(self, total):
        self.weight_sum = float(total)


def make_sampler(entry, rng, seed=0):
    """The roi sampler of one clip for `Workspace.train_sampler`: on the device (cfg.HIP.DEVICE_ROI_SAMPLING, entries made of gt rows only,
    within the kernel's limits) or the host restatement driven by `rng` (the reference's numpy.random stream).  seed: int or callable, see
    DeviceRoiSampler: with a callable the host path consumes NOTHING from `rng` here, i.e. it stays bit-compatible with the reference's stream."""
    from detectandtrack_amd.roi_data import fast_rcnn as frcn_data
    if cfg.HIP.get('DEVICE_ROI_SAMPLING', True) and torch.cuda.is_available():
        try:
            return DeviceRoiSampler(entry, seed=seed)
        except AssertionError:
            pass
    return lambda rois, info: frcn_data.sample_training_blobs(entry, rois, info, rng)
