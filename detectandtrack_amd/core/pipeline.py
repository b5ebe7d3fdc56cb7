"""The execution mode of the inference engine (round 3): several forwards in flight, each replayed as one hipGraph.

The reference's clip loop (lib/core/test_engine.py:124-204) runs one clip at a time: host pre-processing (cv2.resize + mean
subtraction, lib/utils/blob.py:40-90), a synchronous fp32 upload, `model.net`, a host round trip for the detection glue,
`model.keypoint_net`, a host heatmap decode.  Here a "forward" is

    uint8 frames -> pinned staging buffer -> H2D on a copy stream (3 bytes per SOURCE pixel)
                 -> dat_preprocess_frames (the `data` blob, bit-identical to the host path)
                 -> model.net -> dat_box_results -> model.keypoint_net -> heatmap decode      [ONE captured hipGraph]
                 -> one read-back of <= DETECTIONS_PER_IM boxes + 4 x 17T keypoint rows per image

and up to `depth` forwards are in flight on their own HIP streams / blob namespaces (`Workspace.fork`), serviced in COMPLETION
order: the small latency-bound kernels of one forward fill the gaps of another, uploads overlap compute.  A forward carries
`B >= 1` independent images / clips (the N axis of the blobs; every image keeps exactly the results it gets alone, see
core/test.py im_detect_all_batch).  Results are identical to the eager one-clip-at-a-time path: the same kernels with the same
arguments (tests/test_gpu_model.py).

`bench.py` measures this class; `core/test_engine.test_net` runs on it (cfg.HIP.PIPELINE_DEPTH / IMS_PER_FORWARD / CLIP_GRAPH).
"""
import collections
import os
import time

import numpy as np
import torch

from detectandtrack_amd.core.config import cfg
from detectandtrack_amd.core import test as engine
from detectandtrack_amd import workspace as wsmod


class _Slot(object):
    __slots__ = ('ws', 'stream', 'event', 'copy_event', 'graphs', 'pinned', 'dev_u8', 'data')

    def __init__(self, ws):
        self.ws, self.stream = ws, torch.cuda.Stream()
        self.event, self.copy_event = torch.cuda.Event(), torch.cuda.Event()
        self.graphs = collections.OrderedDict()   # geometry key -> ClipGraph, least recently used first (bounded: max_graphs)
        self.pinned = self.dev_u8 = None
        self.data = {}              # geometry key -> the eager path's `data` buffer


class ClipPipeline(object):
    """Runs forwards with up to `depth` of them in flight, each on its own HIP stream + blob namespace, serviced in COMPLETION
    order: a slot is read back and re-used as soon as its forward is done, whichever slot that is (the hardware queues do not
    drain in submission order).  depth=1 is the strictly sequential reference order.  graph=True: every slot replays its forward
    as one captured hipGraph per input geometry (core/clip_graph.py), False: eager launches."""

    def __init__(self, model, ws, depth=4, graph=True, fifo=False, keep_results=True, max_graphs=None):
        assert depth >= 1
        # graphs (and their private memory pools: every activation of a forward) cached per slot, least recently used evicted
        self.max_graphs = max(1, int(cfg.HIP.get('MAX_GRAPHS_PER_SLOT', 6) if max_graphs is None else max_graphs))
        self.graphs_captured = self.graphs_evicted = 0
        self.model, self.depth = model, depth
        self.slots = [_Slot(ws if i == 0 else ws.fork()) for i in range(depth)]
        self.copy_stream = torch.cuda.Stream()
        self.free = list(range(depth))
        self.use_graph = bool(graph)
        self.pending = []           # (slot index, tag, im_info, im_shapes, dev, frames, ClipGraph | None) in submission order
        self.results = []           # (tag, [per-image (cls_boxes, cls_segms, cls_keyps)]) in completion order
        self.keep_results = keep_results
        self.n_det = 0
        self.fifo = bool(fifo)
        self.host_enqueue_s = 0.0
        self.upload_bytes = 0
        self.stage_s = 0.0          # host time in the pageable -> pinned staging copies
        self._stagers = None
        self.host_path_images = 0   # images that took the host glue (exact score ties beyond the device buffers' spare rows)
        self.finish_times = []      # perf_counter() at every completed forward (steady-state rate of a run: see rate())
        self.device_glue = engine.device_results_supported()
        assert self.device_glue, 'the pipelined engine runs the device post-processing path (cfg.HIP.DEVICE_BOX_RESULTS, hard NMS)'

    # ---- slot management ---------------------------------------------------------------------------------------------------
    def _acquire(self):
        """A free slot; with all slots busy: the first pending forward found complete (the oldest one with fifo=True)."""
        if self.free:
            return self.free.pop(0)
        k = 0
        if not self.fifo and len(self.pending) > 1:
            while True:
                done = [j for j, it in enumerate(self.pending) if self.slots[it[0]].event.query()]
                if done:
                    k = done[0]
                    break
                time.sleep(2e-5)
        item = self.pending.pop(k)
        self._finish(item)
        return item[0]

    @staticmethod
    def _geometry(data_shape, im_info, im_shapes):
        return (tuple(int(v) for v in data_shape), tuple(np.asarray(im_info, np.float32).reshape(-1).tolist()),
                tuple(tuple(int(v) for v in sh[:2]) for sh in im_shapes))

    def _enqueue(self, slot, data_dev, im_info, im_shapes, in_place, resident=False):
        """model.net + device glue + keypoint net + decode of one forward on the slot's stream: a graph replay or eager launches.
        in_place: `data_dev` IS the slot's input buffer of this geometry (already filled on the slot's stream).
        resident: `data_dev` is a caller-owned buffer that stays where it is (and filled) for the life of the pipeline: the graph is
        captured reading IT -- one graph per (geometry, buffer) -- instead of a private input buffer that every launch copies into."""
        s = self.slots[slot]
        key = self._geometry(data_dev.shape, im_info, im_shapes)
        if resident:
            key, in_place = key + (int(data_dev.data_ptr()),), True
        if self.use_graph:
            g = s.graphs.get(key)
            if g is None:
                from detectandtrack_amd.core.clip_graph import ClipGraph
                while len(s.graphs) >= self.max_graphs:     # the slot is idle here (acquired): nothing replays the evicted graph
                    _, old = s.graphs.popitem(last=False)
                    del old                                 # frees the graph and its memory pool
                    self.graphs_evicted += 1
                g = s.graphs[key] = ClipGraph(self.model, s.ws, data_dev, im_info, im_shapes, stream=s.stream,
                                              static_data=data_dev if in_place else None)
                self.graphs_captured += 1
            else:
                s.graphs.move_to_end(key)
            return g.launch(data_dev, im_info=im_info, im_shape=im_shapes), g
        with torch.cuda.stream(s.stream):
            prev, wsmod._GLOBAL = wsmod._GLOBAL, s.ws          # the engine functions talk to the global workspace
            try:
                s.ws.FeedBlob('data', data_dev)
                s.ws.FeedBlob('im_info', np.asarray(im_info, np.float32))
                s.ws.RunNet(self.model.net.name)
                scales = [float(v) for v in np.asarray(im_info, np.float32).reshape(-1, 3)[:, 2]]
                return engine.enqueue_results_on_device(self.model, list(im_shapes), scales), None
            finally:
                wsmod._GLOBAL = prev

    # ---- feeding ---------------------------------------------------------------------------------------------------------------
    def submit(self, data_dev, im_info, im_shape, tag=None, resident=False):
        """One forward on a `data` blob that is ALREADY RESIDENT in HBM (fp32 NC[T]HW, B = data_dev.shape[0] images / clips).
        im_info [B, 3]; im_shape: the unscaled (h, w, 3) of the images (one tuple for all, or one per image).  resident=True: the
        caller keeps `data_dev` alive and in place (see _enqueue): no device-to-device copy of the input per forward."""
        slot = self._acquire()
        t0 = time.perf_counter()
        B = int(data_dev.shape[0])
        im_info = np.asarray(im_info, np.float32).reshape(-1, 3)
        assert im_info.shape[0] == B, (im_info.shape, B)
        shapes = [tuple(im_shape)] * B if not isinstance(im_shape[0], (tuple, list)) else [tuple(sh) for sh in im_shape]
        dev, g = self._enqueue(slot, data_dev, im_info, shapes, in_place=False, resident=resident and self.use_graph)
        self.slots[slot].event.record(self.slots[slot].stream)
        self.host_enqueue_s += time.perf_counter() - t0       # host time to enqueue one forward (no synchronisation inside)
        self.pending.append((slot, tag, im_info, shapes, dev, None, g))

    def submit_frames(self, clips, tag=None):
        """One forward from HOST frames: `clips` = B entries, each a list of T uint8 BGR frames (HxWx3 arrays of one size).  The
        frames are staged in the slot's pinned buffer, uploaded on the copy stream as uint8 and prepared on the device."""
        import detectandtrack_amd.utils.blob as blob_utils
        slot = self._acquire()
        s = self.slots[slot]
        t0 = time.perf_counter()
        B, T = len(clips), len(clips[0])
        h, w = clips[0][0].shape[:2]
        n = B * T
        if s.pinned is None or tuple(s.pinned.shape) != (n, h, w, 3):
            s.pinned = torch.empty((n, h, w, 3), dtype=torch.uint8).pin_memory()
            s.dev_u8 = torch.empty((n, h, w, 3), dtype=torch.uint8, device=s.ws.device)
        host = s.pinned.numpy()
        frames = [f for clip in clips for f in clip]
        assert len(frames) == n and all(f.dtype == np.uint8 and f.shape == (h, w, 3) for f in frames), 'clips of one frame size, uint8 HxWx3'
        t1 = time.perf_counter()
        # pageable -> pinned staging copy, frames in parallel (NumPy releases the GIL inside a large copy): a 720p clip is 22 MB, one
        # core moves it in 2-4 ms -- as long as the whole GPU forward
        if self._stagers is None:
            from concurrent.futures import ThreadPoolExecutor
            self._stagers = ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1))
        list(self._stagers.map(lambda kf: np.copyto(host[kf[0]], kf[1]), enumerate(frames)))
        self.stage_s += time.perf_counter() - t1
        with torch.cuda.stream(self.copy_stream):
            s.dev_u8.copy_(s.pinned, non_blocking=True)
            s.copy_event.record(self.copy_stream)
        self.upload_bytes += n * h * w * 3
        scale = blob_utils.test_scale((h, w), cfg.TEST.SCALES[0], cfg.TEST.MAX_SIZE)
        shapes = [(h, w, 3)] * B
        with torch.cuda.stream(s.stream):
            s.stream.wait_event(s.copy_event)
            prev, wsmod._GLOBAL = wsmod._GLOBAL, s.ws
            try:
                gkey = (n, h, w)
                data, _, im_info = blob_utils.frames_to_blob_on_device(s.dev_u8, T, out=s.data.get(gkey))
                s.data[gkey] = data
            finally:
                wsmod._GLOBAL = prev
        dev, g = self._enqueue(slot, s.data[gkey], im_info, shapes, in_place=True)
        s.event.record(s.stream)
        self.host_enqueue_s += time.perf_counter() - t0
        self.pending.append((slot, tag, im_info, shapes, dev, clips, g))

    # ---- completion ------------------------------------------------------------------------------------------------------------
    def _finish(self, item):
        slot, tag, im_info, shapes, dev, clips, g = item
        s = self.slots[slot]
        prev, wsmod._GLOBAL = wsmod._GLOBAL, s.ws
        try:
            with torch.cuda.stream(s.stream):
                res = engine.read_batch_results_from_device(*dev)       # the ONE device -> host transfer of the forward
                out = []
                for i, r in enumerate(res):
                    if r is None:      # exact score ties at the detection limit: this image through the reference's host path
                        if g is not None:   # blob names -> the tensors of the graph that was REPLAYED (not of the last captured one)
                            g.restore_blobs()
                        out.append(self._host_path(s, i, im_info, shapes, clips))
                        continue
                    cls_boxes, cls_keyps = r
                    if cfg.MODEL.KEYPOINTS_ON and sum(len(b) for b in cls_boxes[1:]) == 0:
                        cls_keyps = None
                    out.append((cls_boxes, None, cls_keyps))
        finally:
            wsmod._GLOBAL = prev
        self.n_det = sum(len(b) for b in out[-1][0][1:])
        self.finish_times.append((time.perf_counter(), len(out)))
        if self.keep_results:
            self.results.append((tag, out))

    def _host_path(self, s, i, im_info, shapes, clips):
        """The reference's host glue for ONE image of a finished forward (box_results_with_nms_and_limit keeps every row tied at the
        DETECTIONS_PER_IM cut -- more than the device buffers hold): its rows of the slot's `rois` / `cls_prob` / `bbox_pred` blobs
        go through the host post-processing, its boxes through the keypoint net and the device decode."""
        scales = np.array([im_info[i, 2]])
        ni = int(im_info.shape[0])
        scores, boxes, _ = engine._read_bbox_outputs([np.zeros(shapes[i], np.uint8)], scales, image=i if ni > 1 else None)
        scores, boxes, cls_boxes = engine.box_results_with_nms_and_limit(scores, boxes)
        cls_keyps = None
        if cfg.MODEL.KEYPOINTS_ON and boxes.shape[0] > 0:
            cls_keyps = engine.keypoint_results_on_device(self.model, cls_boxes, boxes, scales, image=i)
        self.host_path_images += 1
        return cls_boxes, None, cls_keyps

    def rate(self, skip=None):
        """Images (clips) per second over the completed forwards, the first `skip` (default: 2 x depth, they include graph capture
        and weight packing) left out; None when too few forwards completed."""
        ft = self.finish_times
        skip = 2 * self.depth if skip is None else skip
        if len(ft) < skip + 2:
            return None
        n = sum(c for _, c in ft[skip + 1:])
        return n / max(ft[-1][0] - ft[skip][0], 1e-9)

    def drain(self):
        """Finish everything in flight (submission order); returns the accumulated (tag, per-image results) list and clears it."""
        while self.pending:
            item = self.pending.pop(0)
            self._finish(item)
            self.free.append(item[0])
        torch.cuda.synchronize()
        out, self.results = self.results, []
        return out
