"""Training input pipeline — counterpart of reference lib/roi_data/loader.py:36-275 (RoIDataLoader).

The reference hides the host cost of a minibatch (frame decoding, ~450 k-anchor RPN labelling, blob packing) behind
worker threads that fill a queue, and enqueuer threads that feed per-GPU blob queues.  Here one process drives one GPU,
so the pipeline is: worker threads -> ordered, bounded queue of ready minibatches -> the training loop.

A worker, for minibatch k (k is global and strictly ordered, so a run is reproducible for any worker count):
  1. asks the clip source for (data, roidb entry, scale) of the k-th clip of the current permutation,
  2. stages `data` through its pinned buffer and copies it to the GPU on the worker's own HIP stream,
  3. labels the anchors — on the device (ops.anchor_overlaps: IoU / arg-max / best-anchor flags, bit-identical to the
     host code) with only the two random sub-samplings on the host, then scatters the <= 256 sampled anchors into the
     dense wide label blobs with dat_scatter_words; or entirely on the host when `device` is None (CPU tests),
  4. synchronises its stream and publishes the minibatch.
The consumer (`get_next_minibatch`) marks the tensors as used by its stream (allocator safety) and feeds the workspace.
"""
import logging
import threading

import numpy as np

from detectandtrack_amd.core.config import cfg
from detectandtrack_amd.roi_data import rpn as rpn_data
from detectandtrack_amd.roi_data import fast_rcnn as frcn_data

logger = logging.getLogger(__name__)


def shuffle_roidb_inds(n, rng, widths=None, heights=None):
    """reference loader.py:96-115: a permutation of the roidb; with TRAIN.ASPECT_GROUPING (and known sizes) landscape and
    portrait clips are paired so that a 2-clip minibatch pads little."""
    if cfg.TRAIN.ASPECT_GROUPING and widths is not None and n % 2 == 0:
        horz = np.asarray(widths) >= np.asarray(heights)
        inds = np.hstack((rng.permutation(np.where(horz)[0]), rng.permutation(np.where(~horz)[0])))
        inds = inds.reshape(-1, 2)
        return inds[rng.permutation(inds.shape[0])].reshape(-1)
    return rng.permutation(n)


class Minibatch(object):
    """One clip ready for the net: blobs name -> numpy array (host mode) or CUDA tensor (device mode)."""
    __slots__ = ('index', 'blobs', 'entry', 'rng', 'sparse', 'label_levels')

    def feed(self, ws):
        """FeedBlob everything and install the Fast R-CNN sampler for the in-net GenerateProposalLabels op."""
        for k, v in self.blobs.items():
            ws.FeedBlob(k, v)
        for name, lvl in self.label_levels.items():         # lets the loss normalise without reading the dense blob back
            ws.blobs[name].host = _WindowCounter(self.sparse, lvl)
        from detectandtrack_amd.roi_data.device_sampler import make_sampler
        # (device kernel by default -- cfg.HIP.DEVICE_ROI_SAMPLING; its draw stream is seeded per minibatch from the minibatch's own RNG, so a
        #  run stays reproducible for any worker count)
        ws.train_sampler = make_sampler(self.entry, self.rng, seed=lambda: int(self.rng.randint(0, 2 ** 31 - 1)))      # (drawn only if the device sampler is built)


class _WindowCounter(object):
    def __init__(self, sparse, level):
        self.sparse, self.level = sparse, level

    def count_in_window(self, h, w):
        return self.sparse.count_in_window(self.level, h, w)


class RoIDataLoader(object):
    """source(i) -> (data float32 (1, 3, T, H, W) mean-subtracted, roidb entry, im_scale); `num_items` clips per epoch."""

    def __init__(self, source, num_items, num_workers=4, queue_size=8, device=None, seed=None, widths=None, heights=None):
        self._source, self._n = source, int(num_items)
        self._device = device
        self._seed = cfg.RNG_SEED if seed is None else seed
        self._sizes = (widths, heights)
        self._cap = max(int(queue_size), 1)
        self._cv = threading.Condition()
        self._ready = {}            # k -> Minibatch | exception
        self._next_k = 0            # next index a worker will take
        self._consumed = 0          # next index the consumer will take
        self._stop = False
        self._perms = {}            # epoch -> permutation
        self._workers = [threading.Thread(target=self._work, name='roi_loader_%d' % i, daemon=True) for i in range(max(num_workers, 1))]
        for t in self._workers:
            t.start()

    # ---- ordering -------------------------------------------------------------------------------------------------------
    def _clip_index(self, k):
        epoch, pos = divmod(k, self._n)
        with self._cv:
            if epoch not in self._perms:
                self._perms[epoch] = shuffle_roidb_inds(self._n, np.random.RandomState(self._seed + 7919 * (epoch + 1)), *self._sizes)
                self._perms.pop(epoch - 2, None)
            return int(self._perms[epoch][pos])

    # ---- worker ---------------------------------------------------------------------------------------------------------
    def _work(self):
        state = {}
        if self._device is not None:
            import torch
            torch.cuda.set_device(self._device)
            state['stream'] = torch.cuda.Stream(device=self._device)
        while True:
            with self._cv:
                while not self._stop and self._next_k >= self._consumed + self._cap:
                    self._cv.wait()
                if self._stop:
                    return
                k = self._next_k
                self._next_k += 1
            try:
                mb = self._build(k, state)
            except Exception as e:  # surfaced to the consumer in order
                logger.exception('minibatch %d failed', k)
                mb = e
            with self._cv:
                self._ready[k] = mb
                self._cv.notify_all()

    def _build(self, k, state):
        data, entry, im_scale = self._source(self._clip_index(k))
        mb = Minibatch()
        mb.index, mb.entry = k, entry
        mb.rng = np.random.RandomState((self._seed + 104729 * (k + 1)) % (2 ** 32))
        if self._device is None:
            mb.sparse, per_level, names, im_info = label_clip_host(entry, im_scale, mb.rng)
            mb.blobs = {'data': data, 'im_info': im_info}
            for lvl_blobs, suffix in zip(per_level, names):
                for name, v in lvl_blobs.items():
                    mb.blobs[name + suffix] = v
        else:
            mb.sparse, mb.blobs, names = label_clip_device(data, entry, im_scale, mb.rng, self._device, state)
        mb.label_levels = {'rpn_labels_int32_wide' + s: i for i, s in enumerate(names)}
        return mb

    # ---- consumer -------------------------------------------------------------------------------------------------------
    def get_next_minibatch(self, timeout=600.0):
        with self._cv:
            k = self._consumed
            if not self._cv.wait_for(lambda: k in self._ready or self._stop, timeout=timeout):
                raise RuntimeError('RoIDataLoader: minibatch %d not ready after %.0f s' % (k, timeout))
            if self._stop and k not in self._ready:
                raise RuntimeError('RoIDataLoader is shut down')
            mb = self._ready.pop(k)
            self._consumed += 1
            self._cv.notify_all()
        if isinstance(mb, Exception):
            raise mb
        if self._device is not None:
            import torch
            cur = torch.cuda.current_stream(self._device)
            for v in mb.blobs.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
        return mb

    def shutdown(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        for t in self._workers:
            t.join(timeout=10.0)


# ---- labelling one clip -------------------------------------------------------------------------------------------------
def _fields_and_gt(entry, im_scale):
    T = entry['boxes'].shape[-1] // 4
    multilevel = cfg.FPN.FPN_ON and cfg.FPN.MULTILEVEL_RPN
    if multilevel:
        foas = rpn_data.fpn_fields(T)
        names = ['_fpn' + str(l) for l in range(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.RPN_MAX_LEVEL + 1)]
    else:
        foas, names = [rpn_data.get_field_of_anchors(cfg.RPN.STRIDE, cfg.RPN.SIZES, cfg.RPN.ASPECT_RATIOS, T)], ['']
    im_h, im_w = np.round(entry['height'] * im_scale), np.round(entry['width'] * im_scale)
    gt = np.where((entry['gt_classes'] > 0) & (entry['is_crowd'] == 0))[0]
    gt_rois = (entry['boxes'][gt] * im_scale).astype(np.float32)
    vis = entry['track_visible'][gt] if 'track_visible' in entry else None
    return T, foas, names, im_h, im_w, gt_rois, vis, np.array([[im_h, im_w, im_scale]], dtype=np.float32)


def label_clip_host(entry, im_scale, rng):
    """rpn.add_rpn_blobs split so that the sparse labels are kept: -> (sparse, per-level dense blobs, suffixes, im_info)."""
    T, foas, names, im_h, im_w, gt_rois, vis, im_info = _fields_and_gt(entry, im_scale)
    stats = rpn_data.anchor_overlap_stats(rpn_data.all_field_anchors(foas), im_h, im_w, gt_rois)
    sparse = rpn_data.sample_rpn_labels(foas, stats, gt_rois, vis, rng)
    return sparse, sparse.dense(), names, im_info


_dev_anchor_cache = {}
_dev_anchor_lock = threading.Lock()


def _device_anchors(foas, device):
    import torch
    key = (tuple(id(f) for f in foas), str(device))
    with _dev_anchor_lock:
        if key not in _dev_anchor_cache:
            _dev_anchor_cache[key] = torch.from_numpy(rpn_data.all_field_anchors(foas)).to(device)
            torch.cuda.synchronize(device)
        return _dev_anchor_cache[key]


def device_overlap_stats(foas, im_h, im_w, gt_rois, device):
    """rpn.anchor_overlap_stats computed by dat_anchor_overlaps on the current stream -> the same four host arrays."""
    import torch
    from detectandtrack_amd.ops import hip_ops as ops
    anchors = _device_anchors(foas, device)
    T = anchors.shape[1] // 4
    gts = torch.from_numpy(np.ascontiguousarray(gt_rois, dtype=np.float32).reshape(-1, 4 * T)).to(device)
    a_max, a_arg, best = ops.anchor_overlaps(anchors, gts, T, float(im_h), float(im_w), float(cfg.TRAIN.RPN_STRADDLE_THRESH))
    a_max, a_arg, best = a_max.cpu().numpy(), a_arg.cpu().numpy(), best.cpu().numpy()
    inside = np.flatnonzero(a_max >= 0)
    has_gt = len(gt_rois) > 0
    return (inside, a_max[inside], a_arg[inside].astype(np.int64), best[inside].astype(bool) if has_gt else np.zeros(len(inside), bool))


def label_clip_device(data, entry, im_scale, rng, device, state):
    """Device mode of one minibatch: everything lands in HBM on the worker's stream; -> (sparse, blobs, suffixes)."""
    import torch
    from detectandtrack_amd.ops import hip_ops as ops
    T, foas, names, im_h, im_w, gt_rois, vis, im_info = _fields_and_gt(entry, im_scale)
    stream = state.get('stream') or torch.cuda.current_stream(device)
    with torch.cuda.stream(stream):
        # clip: pageable -> this worker's pinned staging buffer -> HBM (async on the worker's stream)
        n = data.size
        if state.get('pinned') is None or state['pinned'].numel() < n:
            state['pinned'] = torch.empty(n, dtype=torch.float32).pin_memory()
        staged = state['pinned'][:n].view(data.shape)
        staged.copy_(torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)))
        data_dev = torch.empty(data.shape, dtype=torch.float32, device=device)
        data_dev.copy_(staged, non_blocking=True)
        stats = device_overlap_stats(foas, im_h, im_w, gt_rois, device)
        sparse = rpn_data.sample_rpn_labels(foas, stats, gt_rois, vis, rng)
        offs, vals, views, words = sparse.scatter_plan()
        flat = torch.zeros(words, dtype=torch.int32, device=device)
        blobs = {'data': data_dev, 'im_info': im_info}
        for lvl_views, suffix in zip(views, names):
            for name, (o, shape) in lvl_views.items():
                v = flat[o:o + int(np.prod(shape))]
                if name == 'rpn_labels_int32_wide':
                    v.fill_(-1)
                    blobs[name + suffix] = v.view(shape)
                else:
                    blobs[name + suffix] = v.view(torch.float32).view(shape)
        if len(offs):
            ops.scatter_words(flat, torch.from_numpy(offs).to(device), torch.from_numpy(vals.view(np.int32)).to(device))
        stream.synchronize()
    return sparse, blobs, names
