"""A whole clip as ONE hipGraph launch.

With the glue between the nets on the device (core/test.py enqueue_results_on_device) a clip is a fixed sequence of ~140 kernel
launches with no host decision inside: `model.net`, `dat_box_results`, `model.keypoint_net`, the heatmap decode.  For a fixed input
geometry that sequence is captured once (HIP stream capture through torch.cuda.graph: the executor's torch allocations come from
the graph's private pool, the ctypes launches land on the capturing stream) and replayed per clip: the host cost of a clip drops
from ~140 Python-dispatched launches to one graph launch plus the read-back, and the kernels run back to back.

The reference has no counterpart (Caffe2 runs its nets through a DAG executor with one CUDA stream per op context,
lib/modeling/detector.py:54); results are identical to the eager path by construction -- the same kernels with the same
arguments -- and tests/test_gpu_model.py checks it.
"""
import numpy as np
import torch

from detectandtrack_amd import workspace as wsmod
from detectandtrack_amd.core import test as engine


def ops_ws_generation():
    """How often the C-ABI context's scratch buffer has grown (dat_ws_info).  Captured launches keep the buffer they were captured
    with -- growth retires a buffer instead of freeing it (csrc/c_api.hip dat_ensure_ws) -- so this is bookkeeping for tests."""
    from detectandtrack_amd.ops import hip_ops as ops
    return ops.ws_info()[2]


class ClipGraph(object):
    """model.net + device post-processing (+ keypoint net + decode) for ONE input geometry on one workspace / HIP stream.  The
    geometry -- blob shape, im_info rows (blob height / width / scale per image) and the unscaled image shapes -- is baked into the
    captured launches (`rpn_proposals` / `dat_box_results` take them by value): `launch` refuses anything else."""

    def __init__(self, model, ws, data_like, im_info, im_shape, stream=None, warmup=2, static_data=None, trunk=None):
        """trunk = (n_ops, blob name, static tensor [B*T, h, w, Cs], B, T, C, dtype): the net's per-frame prefix (conv1 ... res2) is NOT part
        of the graph -- the caller fills the static tensor with that prefix's output before every launch (per-frame trunk cache of
        core/pipeline.py) and the captured forward starts at op n_ops; `data_like` then only names the geometry."""
        self.trunk = trunk
        assert engine.device_results_supported(), 'graph capture needs the device-side post-processing (cfg.HIP.DEVICE_BOX_RESULTS)'
        self.model, self.ws = model, ws
        B = int(trunk[3]) if trunk is not None else int(data_like.shape[0])
        # the image scale reaches the device glue as the DOUBLE the reference divides the boxes by (core/test.py:224-231: im_scales from
        # prep_im_for_blob); the `im_info` blob holds its float32 rounding (round 6: the graph path used the rounded one for both)
        self.scales = [float(v) for v in np.asarray(im_info, dtype=np.float64).reshape(-1, 3)[:, 2]]
        self.im_info = np.asarray(im_info, dtype=np.float32).reshape(-1, 3)
        assert self.im_info.shape[0] == B, 'im_info has %d rows for a blob of %d images' % (self.im_info.shape[0], B)
        self.im_shapes = [tuple(im_shape)] * B if not isinstance(im_shape[0], (tuple, list)) else [tuple(sh) for sh in im_shape]
        assert len(self.im_shapes) == B
        self.stream = stream or torch.cuda.current_stream()
        if trunk is not None:
            self.static_data = None
        elif static_data is not None:       # the caller's own input buffer (filled in place before every launch)
            self.static_data = static_data
        else:
            self.static_data = torch.empty_like(data_like)
            self.static_data.copy_(data_like)
        ws.FeedBlob('im_info', self.im_info)            # host -> device copy OUTSIDE the capture (the kernels read the host copy)
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):                     # builds the layers, sizes every workspace, primes the allocator
                self._enqueue()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (torch.distributed's RCCL watchdog polls events) may touch the runtime meanwhile
        with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode='thread_local'):
            self.dev = self._enqueue()
        torch.cuda.synchronize()
        # the blob namespace as THIS graph's replays fill it: a workspace that holds several graphs (one per input geometry) has
        # `ws.blobs` pointing at the tensors of whichever graph was captured last, so whoever reads blobs by name after a replay
        # (the exact-tie host path of core/pipeline.py, FetchBlob in tests) restores this snapshot first
        self.blobs = dict(ws.blobs)
        with torch.cuda.stream(self.stream):
            self.ws_generation = ops_ws_generation()

    def _enqueue(self):
        prev, wsmod._GLOBAL = wsmod._GLOBAL, self.ws    # the engine functions talk to the global workspace
        try:
            if self.trunk is not None:
                self.ws.trunk_ready = self.trunk
            else:
                self.ws.FeedBlob('data', self.static_data)
            self.ws.RunNet(self.model.net.name)
            return engine.enqueue_results_on_device(self.model, self.im_shapes, list(self.scales))
        finally:
            wsmod._GLOBAL = prev

    def launch(self, data_dev, im_info=None, im_shape=None):
        """Enqueue one forward (asynchronous): the resident input is copied into the graph's input buffer on the graph's stream
        (unless it IS that buffer).  im_info / im_shape, when given, must be the captured ones."""
        if self.trunk is not None:          # (the caller has filled the trunk buffer on this stream; nothing to copy)
            with torch.cuda.stream(self.stream):
                self.graph.replay()
            return self.dev
        assert tuple(data_dev.shape) == tuple(self.static_data.shape), \
            'graph captured for a %s blob, launched with %s' % (tuple(self.static_data.shape), tuple(data_dev.shape))
        if im_info is not None:
            assert np.array_equal(np.asarray(im_info, np.float32).reshape(-1, 3), self.im_info), \
                'graph captured for im_info %s, launched with %s' % (self.im_info.tolist(), np.asarray(im_info).tolist())
        if im_shape is not None:
            shapes = [tuple(im_shape)] * len(self.im_shapes) if not isinstance(im_shape[0], (tuple, list)) else [tuple(sh) for sh in im_shape]
            assert [sh[:2] for sh in shapes] == [sh[:2] for sh in self.im_shapes], (shapes, self.im_shapes)
        with torch.cuda.stream(self.stream):
            if data_dev.data_ptr() != self.static_data.data_ptr():
                self.static_data.copy_(data_dev, non_blocking=True)
            self.graph.replay()
        return self.dev

    def restore_blobs(self):
        """Point the workspace's blob names at the tensors this graph's replays write (a copy: eager ops run afterwards may rebind
        names without touching the snapshot)."""
        self.ws.blobs = dict(self.blobs)

    def results(self):
        """Per-image (cls_boxes, cls_keyps) list of the last launched forward (None entries: exact-tie overflow, see
        read_results_from_device); a one-image graph returns its single entry."""
        prev, wsmod._GLOBAL = wsmod._GLOBAL, self.ws
        try:
            with torch.cuda.stream(self.stream):
                res = engine.read_batch_results_from_device(*self.dev)
                return res[0] if len(res) == 1 else res
        finally:
            wsmod._GLOBAL = prev
