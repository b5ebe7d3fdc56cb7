"""NMS dispatch -- mirror of reference lib/core/nms_wrapper.py:29-70, backed by libdat_hip.

Boxes (5 columns) use the cython_nms semantics (suppress at IoU >= thresh, ascending original indices); tubes use
py_cpu_nms_tubes semantics (mean IoU over frames, keep while <= thresh, score order) -- both on the device.  Soft-NMS
(lib/utils/cython_nms.pyx:98-203) is a sequential host algorithm: `dat_soft_nms_host` runs the reference's loop in the same C float
arithmetic.  `nms(..., soft_nms=True)` is rejected loudly: the reference's dispatcher accepts the flag and silently ignores it."""
import ctypes as C

import numpy as np
import torch

from detectandtrack_amd import libdat as L
from detectandtrack_amd.ops import hip_ops as ops

_METHODS = {'hard': 0, 'linear': 1, 'gaussian': 2}


def nms(dets, thresh, soft_nms=False):
    if soft_nms:
        raise ValueError('nms(soft_nms=True): the reference ignores this flag (nms_wrapper.py:49-57); call soft_nms() instead')
    if dets.shape[0] == 0:
        return []
    d = torch.from_numpy(np.ascontiguousarray(dets, dtype=np.float32)).cuda()
    return ops.nms(d, float(thresh)).cpu().numpy().astype(np.int64)


def tube_nms(dets, thresh):
    return nms(dets, thresh)


def soft_nms(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method='linear'):
    """(:29-46) returns what the Cython function returns: (re-scored dets [m, 5], indices into the input [m]).  The reference's
    test.py:766-772 stores that tuple as the class's detections and breaks on the next line; core/test.py unpacks it."""
    if dets.shape[0] == 0:
        return dets, np.zeros((0,), dtype=np.int64)
    if dets.shape[1] > 5:
        raise NotImplementedError('Need to handle tubes..')
    assert method in _METHODS, 'Unknown soft_nms method: {}'.format(method)
    src = np.ascontiguousarray(dets, dtype=np.float32)
    n = src.shape[0]
    out = np.empty((n, 5), dtype=np.float32)
    inds = np.empty((n,), dtype=np.int32)
    m = C.c_int(0)
    rc = L.lib().dat_soft_nms_host(src.ctypes.data_as(C.POINTER(C.c_float)), n, np.float32(sigma), np.float32(overlap_thresh),
                                   np.float32(score_thresh), _METHODS[method], out.ctypes.data_as(C.POINTER(C.c_float)),
                                   inds.ctypes.data_as(C.POINTER(C.c_int)), C.byref(m))
    if rc != L.DAT_OK:
        raise L.DatError('dat_soft_nms_host failed with %d' % rc)
    return out[:m.value], inds[:m.value].astype(np.int64)
