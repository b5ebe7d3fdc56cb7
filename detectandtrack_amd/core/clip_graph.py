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


class ClipGraph(object):
    """model.net + device post-processing (+ keypoint net + decode) for one input geometry on one workspace / HIP stream."""

    def __init__(self, model, ws, data_like, im_info, im_shape, stream=None, warmup=2):
        assert engine.device_results_supported(), 'graph capture needs the device-side post-processing (cfg.HIP.DEVICE_BOX_RESULTS)'
        self.model, self.ws = model, ws
        self.im_info = np.asarray(im_info, dtype=np.float32)
        self.im_shape = tuple(im_shape)
        self.stream = stream or torch.cuda.current_stream()
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

    def _enqueue(self):
        prev, wsmod._GLOBAL = wsmod._GLOBAL, self.ws    # the engine functions talk to the global workspace
        try:
            self.ws.FeedBlob('data', self.static_data)
            self.ws.RunNet(self.model.net.name)
            return engine.enqueue_results_on_device(self.model, self.im_shape, float(self.im_info.reshape(-1)[2]))
        finally:
            wsmod._GLOBAL = prev

    def launch(self, data_dev):
        """Enqueue one clip (asynchronous): the resident clip is copied into the graph's input buffer on the graph's stream."""
        with torch.cuda.stream(self.stream):
            if data_dev.data_ptr() != self.static_data.data_ptr():
                self.static_data.copy_(data_dev, non_blocking=True)
            self.graph.replay()
        return self.dev

    def results(self):
        """(cls_boxes, cls_keyps) of the last launched clip, or None on the exact-tie overflow (see read_results_from_device)."""
        prev, wsmod._GLOBAL = wsmod._GLOBAL, self.ws
        try:
            with torch.cuda.stream(self.stream):
                return engine.read_results_from_device(*self.dev)
        finally:
            wsmod._GLOBAL = prev
