"""NMS dispatch — mirror of reference lib/core/nms_wrapper.py:49-70, backed by the device kernels.

Boxes (5 columns) use the cython_nms semantics (suppress at IoU >= thresh, ascending original indices); tubes
use py_cpu_nms_tubes semantics (mean IoU over frames, keep while <= thresh, score order)."""
import numpy as np
import torch

from detectandtrack_amd.ops import hip_ops as ops


def nms(dets, thresh, soft_nms=False):
    if dets.shape[0] == 0:
        return []
    d = torch.from_numpy(np.ascontiguousarray(dets, dtype=np.float32)).cuda()
    return ops.nms(d, float(thresh)).cpu().numpy().astype(np.int64)


def tube_nms(dets, thresh):
    return nms(dets, thresh)


def soft_nms(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method='linear'):
    raise NotImplementedError('Soft-NMS is disabled in every shipped config (TEST.SOFT_NMS.ENABLED False)')
