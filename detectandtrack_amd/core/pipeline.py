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


def _cu_masked_stream(spec):
    """EXPERIMENT (DAT_SLOT_CUS): a HIP stream restricted to a CU subset (hipExtStreamCreateWithCUMask), wrapped for torch.
    spec: 'lo-hi[+lo-hi...]' bit ranges of the CU mask (256 CUs on MI355X; the driver spreads mask bits round-robin over the 8 XCDs)."""
    import ctypes
    bits = 0
    for part in spec.split('+'):
        lo, hi = [int(v) for v in part.split('-')]
        for b in range(lo, hi + 1):
            bits |= 1 << b
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    hip = ctypes.CDLL('libamdhip64.so')
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, 'hipExtStreamCreateWithCUMask failed: %d' % rc
    return torch.cuda.ExternalStream(st.value)


def _set_share(stream, percent):
    from detectandtrack_amd.ops import hip_ops as ops
    with torch.cuda.stream(stream):
        ops.persistent_share(max(1, min(100, int(percent))))


class _Slot(object):
    __slots__ = ('ws', 'stream', 'event', 'copy_event', 'graphs', 'pinned', 'dev_u8', 'u8', 'data', 'live', 'gather_event')
    _made = 0

    def __init__(self, ws):
        spec = os.environ.get('DAT_SLOT_CUS', '')          # e.g. '0-127,128-255': slot i takes entry i % n ('all' = an ordinary stream)
        parts = [p for p in spec.split(',') if p]
        mine = parts[_Slot._made % len(parts)] if parts else 'all'
        _Slot._made += 1
        self.ws, self.stream = ws, (torch.cuda.Stream() if mine == 'all' else _cu_masked_stream(mine))
        self.event, self.copy_event = torch.cuda.Event(), torch.cuda.Event()
        self.graphs = collections.OrderedDict()   # geometry key -> ClipGraph, least recently used first (bounded: max_graphs)
        self.pinned = self.dev_u8 = None
        self.u8 = {}                # geometry key (frames, h, w) -> (pinned staging tensor, device uint8 tensor)
        self.data = {}              # geometry key -> the forward's input: the fp32 `data` buffer, or the FrameBlob over the slot's uint8 buffer
        self.live = {}              # (frame-trunk cache) geometry key -> this slot's static buffer of gathered prefix outputs [B*T, h, w, Cs]
        self.gather_event = None    # recorded after the slot's last gather out of the trunk pool


class FrameTrunkCache(object):
    """Per-frame cache of the net's frame-wise prefix (conv1 / pool1 / res2: time kernel 1, ResNet3D.py:258-275) for the pipelined engine
    (round 5; cfg.HIP.FRAME_TRUNK_CACHE frames; the eager engine's version lives in workspace.Executor._run_cached_trunk).

    The reference scores a video with one clip per key frame, stride 1 (lib/utils/video.py:149-201): consecutive clips share T - 1 of T
    frames and everything is recomputed.  Here a forward of B clips uploads, pre-processes and runs the prefix for its NEW frames only --
    on a dedicated trunk stream with its own blob namespace, eager launches --, stores their outputs in slots of one pool tensor
    [capacity, h, w, Cs] (dat_copy_frames), and every slot of the pipeline gathers its B * T frames from the pool into the static input
    of its captured hipGraph, which starts at the first op AFTER the prefix.  Ordering: a forward's gather waits for the trunk event of
    its own submit; the trunk stream waits for every slot's last gather before it overwrites pool slots (least recently used first,
    never a frame of the forward being assembled)."""

    def __init__(self, model, ws, capacity):
        from detectandtrack_amd.workspace import Executor
        self.model, self.capacity = model, int(capacity)
        self.n_ops, self.live = Executor.trunk_split(model.net)
        assert self.n_ops > 0, 'frame-trunk cache: this net has no per-frame prefix'
        self.ws = ws.fork()
        self.stream = torch.cuda.Stream()
        self.event = torch.cuda.Event()
        self.copy_event = torch.cuda.Event()
        self.pool = None                        # [capacity, h, w, Cs] in the activation dtype
        self.meta = None                        # (C, dtype) of the prefix output
        self.slot_of = collections.OrderedDict()   # frame id -> pool slot, least recently used first
        self.free = list(range(self.capacity))
        self.pinned = self.dev_u8 = None
        self.frames_computed = self.frames_requested = 0
        self._hw = self._info = None
        self.resets = 0

    def reset(self, capacity=None):
        """Forget every cached frame (and the pool: the next forward allocates one for ITS geometry).  The caller has finished every forward
        in flight -- they read the pool or a buffer gathered from it.  PoseTrack videos differ in resolution: a frame of another size starts
        a new pool (frames of the previous video are not asked for again: clips never cross a video, utils/video.py:149-201)."""
        self.event.synchronize()
        if capacity is not None:
            self.capacity = int(capacity)
        self.pool = self.meta = None
        self._hw = self._info = None
        self.slot_of.clear()
        self.free = list(range(self.capacity))
        self.resets += 1

    def _run_prefix(self, data, im_info):
        """the prefix ops on `data` [1, 3, n, H, W] (n frames as one clip: the ops are frame-wise) -> the output blob"""
        from detectandtrack_amd.workspace import Executor
        ws = self.ws
        ws.FeedBlob('data', data)
        ws.FeedBlob('im_info', np.asarray(im_info, np.float32))
        ex = Executor(ws, self.model.net)
        ex._plan_rpn_siblings()
        ex._plan_keyframe_dce()
        for i in range(self.n_ops):
            if i not in ex._skip:
                getattr(ex, 'op_' + self.model.net.ops[i].type)(i, self.model.net.ops[i])
        return ws.blobs[self.live]

    def assemble(self, pipe, frames_by_id, ids, T, wait_events):
        """ids: the B * T frame ids of a forward (clip-major); frames_by_id: id -> uint8 HxWx3 frame for (at least) the ids that are not
        cached.  Uploads + computes the missing frames on the trunk stream and returns (pool slots of `ids`, im_info rows, geometry)."""
        import detectandtrack_amd.utils.blob as blob_utils
        from detectandtrack_amd.ops import hip_ops as ops
        self.frames_requested += len(ids)
        need = set(ids)
        assert len(need) <= self.capacity, 'cfg.HIP.FRAME_TRUNK_CACHE %d < the %d distinct frames of one forward' % (self.capacity, len(need))
        new = []
        for fid in ids:
            if fid not in self.slot_of and fid not in new:
                new.append(fid)
        for fid in ids:
            if fid in self.slot_of:
                self.slot_of.move_to_end(fid)
        h, w = frames_by_id[ids[0]].shape[:2] if ids[0] in frames_by_id else self._hw
        self._hw = (h, w)
        if new:
            n = len(new)
            while len(self.free) < n:           # evict least recently used frames that this forward does not use
                victim = next(f for f in self.slot_of if f not in need)
                self.free.append(self.slot_of.pop(victim))
            # (the previous upload out of the pinned buffer must have left it: its copy event)
            self.copy_event.synchronize()
            if self.pinned is None or self.pinned.shape[0] < n or tuple(self.pinned.shape[1:]) != (h, w, 3):
                # a larger staging pair: the trunk stream may still be reading the old device buffer, and dropping it hands the memory back
                # to the allocator of THIS thread's stream, which knows nothing about that reader -- wait for the prefix that read it
                self.event.synchronize()
                cap = max(n, 8)
                self.pinned = torch.empty((cap, h, w, 3), dtype=torch.uint8).pin_memory()
                self.dev_u8 = torch.empty((cap, h, w, 3), dtype=torch.uint8, device=self.ws.device)
            host = self.pinned.numpy()
            for k, fid in enumerate(new):
                f = frames_by_id[fid]
                assert f.dtype == np.uint8 and f.shape == (h, w, 3), 'frames of one size, uint8 HxWx3'
                np.copyto(host[k], f)
            with torch.cuda.stream(pipe.copy_stream):
                pipe.copy_stream.wait_event(self.event)         # the trunk stream is done reading the previous upload
                self.dev_u8[:n].copy_(self.pinned[:n], non_blocking=True)
                self.copy_event.record(pipe.copy_stream)
            pipe.upload_bytes += n * h * w * 3
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(self.copy_event)
                for ev in wait_events:                          # gathers that still read pool slots about to be overwritten
                    self.stream.wait_event(ev)
                prev, wsmod._GLOBAL = wsmod._GLOBAL, self.ws
                try:
                    # (lazy: with cfg.HIP.STEM_FROM_UINT8 the fused stem of the prefix reads the uploaded frames themselves)
                    data, _, im_info = blob_utils.frames_to_blob_on_device(self.dev_u8[:n], n, lazy=True)
                    out = self._run_prefix(data, im_info[:1])
                finally:
                    wsmod._GLOBAL = prev
                assert out.t.shape[0] == n
                assert self.pool is None or tuple(out.t.shape[1:]) == tuple(self.pool.shape[1:]), \
                    'frame-trunk cache: frames of another size (%s vs %s): one geometry per pipeline' % (tuple(out.t.shape[1:]), tuple(self.pool.shape[1:]))
                if self.pool is None:
                    self.pool = torch.empty((self.capacity,) + tuple(out.t.shape[1:]), dtype=out.t.dtype, device=out.t.device)
                    self.meta = (out.C, out.dt)
                    self._info = (float(im_info[0, 0]), float(im_info[0, 1]), float(im_info[0, 2]))
                slots = [self.free.pop() for _ in new]
                ops.copy_frames(out.t, list(range(n)), self.pool, slots)
                self.event.record(self.stream)
            for fid, sl in zip(new, slots):
                self.slot_of[fid] = sl
            self.frames_computed += n
        B = len(ids) // T
        im_info = np.tile(np.array([self._info], dtype=np.float64), (B, 1))
        return [self.slot_of[fid] for fid in ids], im_info


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
        # several forwards in flight: their persistent HBM-bound conv kernels take a share of the CUs (cfg.HIP.PERSISTENT_CU_SHARE) so that
        # the other forwards' MFMA-bound kernels run beside them; a lone forward keeps the whole chip.  (Set on every slot's own context:
        # pooled stream handles recur, so depth 1 resets it.)
        self.persistent_share = int(os.environ.get('DAT_PERSISTENT_CU_SHARE', cfg.HIP.get('PERSISTENT_CU_SHARE', 100))) if depth >= 2 else 100
        for s_ in self.slots:
            _set_share(s_.stream, self.persistent_share)
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
        self.rerun_images = 0       # images whose detections tied beyond the device buffers' spare rows: device glue re-run with more rows
        self.host_path_images = 0   # images that took the reference's host glue (only if that re-run still overflowed)
        self.finish_times = []      # perf_counter() at every completed forward (steady-state rate of a run: see rate())
        self.trunk = None           # FrameTrunkCache, made by the first submit_frames(..., frame_ids=...) with cfg.HIP.FRAME_TRUNK_CACHE > 0
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

    def _enqueue(self, slot, data_dev, im_info, im_shapes, in_place, resident=False, trunk=None):
        """model.net + device glue + keypoint net + decode of one forward on the slot's stream: a graph replay or eager launches.
        in_place: `data_dev` IS the slot's input buffer of this geometry (already filled on the slot's stream).
        resident: `data_dev` is a caller-owned buffer that stays where it is (and filled) for the life of the pipeline: the graph is
        captured reading IT -- one graph per (geometry, buffer) -- instead of a private input buffer that every launch copies into."""
        s = self.slots[slot]
        key = self._geometry(data_dev.shape, im_info, im_shapes)
        if resident:
            key, in_place = key + (int(data_dev.data_ptr()),), True
        if trunk is not None:       # (the graph starts behind the per-frame prefix and reads the slot's gathered buffer: its own geometry key)
            key = key + ('trunk',)
        if self.use_graph:
            g = s.graphs.get(key)
            if g is None:
                from detectandtrack_amd.core.clip_graph import ClipGraph
                while len(s.graphs) >= self.max_graphs:     # the slot is idle here (acquired): nothing replays the evicted graph
                    _, old = s.graphs.popitem(last=False)
                    del old                                 # frees the graph and its memory pool
                    self.graphs_evicted += 1
                g = s.graphs[key] = ClipGraph(self.model, s.ws, data_dev, im_info, im_shapes, stream=s.stream,
                                              static_data=data_dev if in_place else None, trunk=trunk)
                self.graphs_captured += 1
            else:
                s.graphs.move_to_end(key)
            return g.launch(data_dev, im_info=im_info, im_shape=im_shapes), g
        with torch.cuda.stream(s.stream):
            prev, wsmod._GLOBAL = wsmod._GLOBAL, s.ws          # the engine functions talk to the global workspace
            try:
                if trunk is not None:
                    s.ws.trunk_ready = trunk
                else:
                    s.ws.FeedBlob('data', data_dev)
                s.ws.FeedBlob('im_info', np.asarray(im_info, np.float32))
                s.ws.RunNet(self.model.net.name)
                scales = [float(v) for v in np.asarray(im_info, np.float64).reshape(-1, 3)[:, 2]]      # (the exact double, see ClipGraph)
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
        im_info = np.asarray(im_info, np.float64).reshape(-1, 3)        # (float64: the scale column is used as the reference's double)
        assert im_info.shape[0] == B, (im_info.shape, B)
        shapes = [tuple(im_shape)] * B if not isinstance(im_shape[0], (tuple, list)) else [tuple(sh) for sh in im_shape]
        dev, g = self._enqueue(slot, data_dev, im_info, shapes, in_place=False, resident=resident and self.use_graph)
        self.slots[slot].event.record(self.slots[slot].stream)
        self.host_enqueue_s += time.perf_counter() - t0       # host time to enqueue one forward (no synchronisation inside)
        self.pending.append((slot, tag, im_info, shapes, dev, None, g))

    def _submit_frames_cached(self, clips, tag, frame_ids):
        """submit_frames through the per-frame trunk cache: only the frames no earlier forward has seen are uploaded and run through
        conv1 ... res2; the forward itself is the graph behind that prefix on the slot's gathered buffer."""
        from detectandtrack_amd.ops import hip_ops as ops
        B, T = len(clips), len(clips[0])
        h, w = clips[0][0].shape[:2]
        # capacity: the configured frame count, but never less than two forwards' worth of frames (a forward must fit next to the frames the
        # forwards in flight still gather; ADVICE r5: was a hard assert inside assemble)
        want = max(int(cfg.HIP.FRAME_TRUNK_CACHE), 2 * B * T)
        if self.trunk is None:
            self.trunk = FrameTrunkCache(self.model, self.slots[0].ws, want)
            _set_share(self.trunk.stream, self.persistent_share)     # (the prefix runs beside the slots' forwards)
        tr = self.trunk
        if (tr.pool is not None and (h, w) != tr._hw) or want > tr.capacity:
            # frames of another size (the next video) or a larger forward: one pool holds one geometry.  Finish what is in flight (results are
            # kept), then start over with an empty cache -- mixed-resolution datasets run through the cached engine (before: an assert
            # AFTER the upload and the prefix had run)
            self._finish_pending()
            tr.reset(capacity=max(want, tr.capacity))
        slot = self._acquire()
        s = self.slots[slot]
        t0 = time.perf_counter()
        ids = [tuple(f) if isinstance(f, list) else f for clip_ids in frame_ids for f in clip_ids]
        assert len(ids) == B * T, 'frame_ids: one id per frame of every clip'
        frames_by_id = {}
        for clip, clip_ids in zip(clips, frame_ids):
            for f, fid in zip(clip, clip_ids):
                frames_by_id.setdefault(tuple(fid) if isinstance(fid, list) else fid, f)
        slots_idx, im_info = tr.assemble(self, frames_by_id, ids, T, [sl.gather_event for sl in self.slots if sl.gather_event is not None])
        shapes = [(h, w, 3)] * B
        gkey = (B * T,) + tuple(tr.pool.shape[1:])
        with torch.cuda.stream(s.stream):
            s.stream.wait_event(tr.event)
            live = s.live.get(gkey)
            if live is None:
                live = s.live[gkey] = torch.empty((B * T,) + tuple(tr.pool.shape[1:]), dtype=tr.pool.dtype, device=tr.pool.device)
            ops.copy_frames(tr.pool, slots_idx, live, list(range(B * T)))
            if s.gather_event is None:
                s.gather_event = torch.cuda.Event()
            s.gather_event.record(s.stream)
        trunk = (tr.n_ops, tr.live, live, B, T, tr.meta[0], tr.meta[1])
        geom = torch.empty((B, 3, T, int(im_info[0, 0]), int(im_info[0, 1])), device='meta')      # (names the geometry of the forward: no memory)
        dev, g = self._enqueue(slot, geom, im_info, shapes, in_place=True, trunk=trunk)
        s.event.record(s.stream)
        self.host_enqueue_s += time.perf_counter() - t0
        self.pending.append((slot, tag, im_info, shapes, dev, clips, g))

    def submit_frames(self, clips, tag=None, frame_ids=None):
        """One forward from HOST frames: `clips` = B entries, each a list of T uint8 BGR frames (HxWx3 arrays of one size).  The
        frames are staged in the slot's pinned buffer, uploaded on the copy stream as uint8 and prepared on the device.
        frame_ids (B lists of T hashable ids, e.g. (video, frame number)) with cfg.HIP.FRAME_TRUNK_CACHE > 0: frames with an id seen
        before are neither uploaded nor run through the net's per-frame prefix again (FrameTrunkCache)."""
        import detectandtrack_amd.utils.blob as blob_utils
        if frame_ids is not None and int(cfg.HIP.FRAME_TRUNK_CACHE) > 0 and cfg.MODEL.VIDEO_ON:
            return self._submit_frames_cached(clips, tag, frame_ids)
        slot = self._acquire()
        s = self.slots[slot]
        t0 = time.perf_counter()
        B, T = len(clips), len(clips[0])
        h, w = clips[0][0].shape[:2]
        n = B * T
        gkey = (n, h, w)
        if gkey not in s.u8:
            # one staging pair per input geometry: a captured graph reads ITS device buffer (the frames are the forward's input since
            # round 6 -- the fused stem evaluates the pre-processing itself --, so the buffer a graph was captured on must stay where it is)
            s.u8[gkey] = (torch.empty((n, h, w, 3), dtype=torch.uint8).pin_memory(), torch.empty((n, h, w, 3), dtype=torch.uint8, device=s.ws.device))
        s.pinned, s.dev_u8 = s.u8[gkey]
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
                data, _, im_info = blob_utils.frames_to_blob_on_device(s.dev_u8, T, out=s.data.get(gkey) if torch.is_tensor(s.data.get(gkey)) else None,
                                                                     lazy=True)
                if not torch.is_tensor(data) and gkey in s.data and not torch.is_tensor(s.data[gkey]):
                    data = s.data[gkey]         # the SAME FrameBlob object for a geometry: the graph was captured on it
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
                if any(r is None for r in res):
                    # exact score ties at the detection limit kept more rows than the device buffers hold (the limit rule keeps EVERY score
                    # tied with the DETECTIONS_PER_IM-th best, core/test.py:795-800): run the device glue + keypoint net again, eagerly, with
                    # as many rows per image as the rule asked for -- the same kernels, the reference's result, no host post-processing
                    if g is not None:       # blob names -> the tensors of the graph that was REPLAYED (not of the last captured one)
                        g.restore_blobs()
                    res = self._rerun_with_all_tied_rows(res, dev, im_info, shapes)
                out = []
                for i, r in enumerate(res):
                    if r is None:      # (not expected after the re-run; kept as the reference's own glue)
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

    def _rerun_with_all_tied_rows(self, res, dev, im_info, shapes):
        """Only the images that overflowed are recomputed: their row segment of the forward's blobs through dat_box_results with
        n_out[i, 1] rows, their boxes through the keypoint net and the decode."""
        n = dev[1].cpu().numpy().reshape(-1, 2)
        info = np.asarray(im_info, np.float64).reshape(-1, 3)
        out = list(res)
        for i, r in enumerate(res):
            if r is not None:
                continue
            again = engine.enqueue_results_on_device(self.model, tuple(shapes[i]), float(info[i, 2]), out_cap=int(n[i, 1]),
                                                     image=i if len(res) > 1 else None)
            out[i] = engine.read_results_from_device(*again)
            self.rerun_images += 1
        return out

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

    def _finish_pending(self):
        """Finish everything in flight (submission order); the results stay in `self.results`."""
        while self.pending:
            item = self.pending.pop(0)
            self._finish(item)
            self.free.append(item[0])

    def drain(self):
        """Finish everything in flight (submission order); returns the accumulated (tag, per-image results) list and clears it."""
        self._finish_pending()
        torch.cuda.synchronize()
        out, self.results = self.results, []
        return out
