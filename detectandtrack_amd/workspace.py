"""Workspace + executor: the `caffe2.python.workspace` surface the reference engines use
(`FeedBlob` / `RunNet` / `FetchBlob` / `CreateNet`, lib/core/test.py:190-232,611-621), running the recorded op
lists of `modeling.detector` as HIP kernel launches through the C ABI (detectandtrack_amd.ops.hip_ops).

Blobs live on the MI355X as NDHWC tensors (fp32 parity mode / bf16 performance mode, cfg.HIP.DTYPE); `FetchBlob`
converts back to the reference's NC[T]HW fp32 layout, so engine code written against the reference reads the
same arrays.  One process drives one GPU (reference inference model: one subprocess per GPU,
lib/utils/subprocess.py:38-63).  There is no CPU execution path.
"""
import collections
import logging

import numpy as np
import torch

from detectandtrack_amd.core.config import cfg
from detectandtrack_amd.ops import hip_ops as ops

logger = logging.getLogger(__name__)


class Blob(object):
    """A device blob.  kind: 'fmap' [N*T,H,W,Cs] | 'rows' [1,1,R,Cs] (FC activations) | 'mat' fp32 tensor |
    'rois' fp32 [cap, cols] + device count."""
    __slots__ = ('t', 'kind', 'N', 'T', 'C', 'dt', 'five_d', 'count', 'sigmoid_of', 'host', 'keyframe', 't2c', 'split', 'tsel')

    def __init__(self, t, kind, N=1, T=1, C=0, dt=0, five_d=False, count=None):
        self.t, self.kind, self.N, self.T, self.C, self.dt = t, kind, N, T, C, dt
        self.five_d = five_d
        self.count = count
        self.sigmoid_of = None
        self.host = None
        self.split = None       # bf16x3 mode: the hi / lo bf16 split of `t`, made by the first conv that reads the blob and shared by the others
        self.tsel = None        # (k, T): a LAZY SliceKeyFrame -- `t` still holds all T frames of every clip, the blob is frame k of each (no copy:
                                # the reading conv / RoIAlign address that frame themselves)
        self.t2c = False        # time moved into channels (detector.py:480-491): still stored as T frames of C channels
        self.keyframe = None    # set when only this frame of a T-frame blob was computed (cfg.HIP.KEYFRAME_DCE)


class Workspace(object):
    def __init__(self, device=0, dtype=None):
        self.device = torch.device('cuda', device)
        self.dtype = dtype      # 'bf16' | 'fp32' arithmetic of THIS workspace; None = cfg.HIP.DTYPE at run time
        self.blobs = {}
        self.params = {}        # name -> np.ndarray (host master copy, reference blob layout)
        self.nets = {}
        self._layers = {}       # (net name, op index) -> prepared ConvLayer etc.
        self._dev_params = {}
        self.conv_log = None    # when a list: (name, algorithmic flops) per conv launch (bench roofline leg)
        self.train_sampler = None   # training: callable(rois, im_info) -> sampled Fast R-CNN blobs (training.py)
        self.trunk_cache = collections.OrderedDict()   # frame id -> (per-frame trunk output, C, dtype)  (cfg.HIP.FRAME_TRUNK_CACHE)
        self.trunk_request = None   # (frame ids of the next clip, ids of the frames in the fed `data` blob)
        self.trunk_ready = None     # (first op to run, blob name, tensor [N*T,h,w,Cs], N, T, C, dtype): the prefix's output, assembled by the caller

    # ---- reference workspace API -------------------------------------------------------------------------------
    def FeedBlob(self, name, arr):
        name = _unscoped(name)
        if getattr(arr, 'materialise', None) is not None and hasattr(arr, 'frames'):
            # utils.blob.FrameBlob: the `data` blob described by the uploaded uint8 frames (the fused stem reads them directly)
            self.blobs[name] = Blob(arr, 'u8frames', N=arr.N, T=arr.T, C=3, five_d=arr.five_d)
            return
        if isinstance(arr, torch.Tensor):
            t = arr.to(self.device)
        else:
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
        b = Blob(t, 'mat')
        b.host = None if isinstance(arr, torch.Tensor) else np.asarray(arr)
        self.blobs[name] = b

    def FetchBlob(self, name):
        name = _unscoped(name)
        if name in self.params and name not in self.blobs:
            return self.params[name]
        b = self.blobs[name]
        if b.kind == 'u8frames':        # (never materialised on the hot path: the blob as dat_preprocess_frames writes it)
            return b.t.materialise().cpu().numpy()
        if b.kind == 'fmap':
            t = b.t
            if b.tsel is not None:      # lazy key-frame slice: pick frame k of every clip now
                k, tf = b.tsel
                t = t.view((b.N, tf) + tuple(t.shape[1:]))[:, k].contiguous()
            out = ops.to_ncdhw(t, b.dt, b.N, b.C, b.T).cpu().numpy()
            if b.t2c:   # (N, C, T, H, W) -> (N, T*C, H, W), channel = t*C + c
                return np.ascontiguousarray(out.transpose(0, 2, 1, 3, 4)).reshape(b.N, b.T * b.C, out.shape[3], out.shape[4])
            return out if b.five_d else out[:, :, 0]
        if b.kind == 'rows':
            r = b.t.shape[2]
            m = ops.to_ncdhw(b.t.view(r, 1, 1, b.t.shape[3]), b.dt, r, b.C, 1).cpu().numpy().reshape(r, b.C)
            return _valid_rows(b, m)
        if b.kind == 'rois':
            return _valid_rows(b, b.t.cpu().numpy())
        arr = b.t.cpu().numpy()
        if b.sigmoid_of is not None:
            arr = 1.0 / (1.0 + np.exp(-arr))
        return _valid_rows(b, arr)

    def HasBlob(self, name):
        return _unscoped(name) in self.blobs or _unscoped(name) in self.params

    def Blobs(self):
        return list(self.blobs.keys())

    # decisions an Executor derives from "which ops of the REGISTERED nets read this blob" and caches in `_layers`
    _READER_KEYS = ('slice_lazy', 'pad_unread', 'conv_reader')

    def CreateNet(self, net):
        self.nets[net.name] = net
        # a new net may read a blob that so far only pad-blind / frame-addressing ops read: the cached reader scans (lazy SliceKeyFrame,
        # skipped zero fill of the padding channels, epilogue-written bf16x3 splits) are re-derived on the next forward.  `_layers` is
        # shared with the forks, so it is edited in place.
        for k in [k for k in self._layers if isinstance(k, tuple) and k and k[0] in self._READER_KEYS]:
            del self._layers[k]
        return net

    def RunNet(self, name):
        name = getattr(name, 'name', name)
        net = self.nets[name]
        Executor(self, net).run()

    def ResetWorkspace(self):
        self.blobs.clear()
        self._layers.clear()
        self._dev_params.clear()
        self.trunk_cache.clear()
        self.trunk_request = None

    def trunk_missing(self, frame_ids):
        """Frame ids of a clip whose per-frame trunk output is not cached (first occurrences, in clip order)."""
        seen, out = set(), []
        for fid in frame_ids:
            if fid not in self.trunk_cache and fid not in seen:
                seen.add(fid)
                out.append(fid)
        return out

    def fork(self):
        """A second blob namespace over the SAME parameters / packed layers / nets: lets two clips be in flight on
        two HIP streams (host post-processing of clip i overlaps the device forward of clip i+1)."""
        w = Workspace.__new__(Workspace)
        w.device = self.device
        w.dtype = self.dtype
        w.blobs = {}
        w.params, w.nets, w._layers, w._dev_params = self.params, self.nets, self._layers, self._dev_params
        w.conv_log = None
        w.train_sampler = self.train_sampler
        w.trunk_cache, w.trunk_request, w.trunk_ready = self.trunk_cache, None, None
        return w

    # ---- parameters -----------------------------------------------------------------------------------------------
    def dev_param(self, name):
        if name not in self._dev_params:
            self._dev_params[name] = torch.from_numpy(np.ascontiguousarray(self.params[name], dtype=np.float32)).to(self.device)
        return self._dev_params[name]

    def set_param(self, name, arr):
        self.params[name] = np.asarray(arr, dtype=np.float32)
        dev = self._dev_params.get(name)
        if dev is not None and tuple(dev.shape) == tuple(self.params[name].shape):
            # a device master exists (it may be a slice of a Trainer's flat buffer that the SGD update and the packed layers alias):
            # overwrite it IN PLACE so that every alias keeps pointing at the live weights
            dev.copy_(torch.from_numpy(self.params[name]).to(dev.device))
        else:
            self._dev_params.pop(name, None)
        self._layers.clear()
        self.param_epoch = getattr(self, 'param_epoch', 0) + 1      # bumped whenever packed layers were invalidated

    def params_from_device(self, names=None):
        """Copy the device fp32 masters (updated in place by training) back into the host parameter dict (checkpoints)."""
        for n in (names if names is not None else list(self._dev_params)):
            if n in self._dev_params:
                self.params[n] = self._dev_params[n].cpu().numpy()


def blob_as_matrix(b):
    """A head output as a device fp32 matrix [rows, C]: 'mat' blobs as they are, FC 'rows' blobs ([1,1,R,Cs] in the activation
    dtype) converted on the device."""
    if b.kind == 'mat':
        return b.t
    assert b.kind == 'rows', b.kind
    r, cs = b.t.shape[2], b.t.shape[3]
    return ops.to_ncdhw(b.t.view(r, 1, 1, cs), b.dt, r, b.C, 1).view(r, b.C)


def _unscoped(name):
    name = str(name)
    return name[name.rfind('/') + 1:]  # 'gpu_0/rois' -> 'rois' (core.ScopedName, lib/utils/c2.py)


def _count(b):
    return int(b.count.item()) if isinstance(b.count, torch.Tensor) else int(b.count)


def _valid_rows(b, arr):
    """The live rows of a row-counted blob: arr[:count], or -- several images per forward, count = one entry per image over equal
    row segments -- the images' live rows concatenated in image order (what the reference's batched blobs hold, col 0 = image)."""
    if b.count is None:
        return arr
    if isinstance(b.count, torch.Tensor) and b.count.numel() > 1:
        cnt = b.count.cpu().numpy().reshape(-1)
        if arr.ndim == 3 and arr.shape[0] == len(cnt):      # [n_images, rows, cols] view of a per-level proposal blob
            arr = arr.reshape(-1, arr.shape[-1])
        seg = arr.shape[0] // len(cnt)
        assert seg * len(cnt) == arr.shape[0], (arr.shape, len(cnt))
        return np.concatenate([arr[i * seg:i * seg + int(c)] for i, c in enumerate(cnt)], axis=0)
    return arr[:_count(b)]


def _mode(ws=None):
    mode = (getattr(ws, 'dtype', None) if ws is not None else None) or cfg.HIP.DTYPE
    assert mode in ('bf16', 'fp16', 'fp32', 'bf16x3'), 'cfg.HIP.DTYPE %r: bf16 | fp16 | fp32 | bf16x3' % (mode,)
    if mode in ('bf16', 'fp16', 'bf16x3'):
        # the 16-bit format is a property of the loaded library build (one per process: DAT_H16 selects libdat_hip.so / libdat_hip_f16.so)
        want = 'fp16' if mode == 'fp16' else 'bf16'
        assert ops.L.H16 == want, ('cfg.HIP.DTYPE %r needs the %s build of the library: start the process with DAT_H16=%s (loaded: %s)'
                                   % (mode, want, want, ops.L.H16))
    return mode


def _dt(ws=None):
    """element type of the activations: bf16 in the performance mode; fp32 in the parity mode AND in 'bf16x3', whose convs split
    their fp32 operands into bf16 hi / lo parts on the fly (ops.ConvLayer x3)"""
    return ops.BF16 if _mode(ws) in ('bf16', 'fp16') else ops.F32       # (ops.BF16 = the library's 16-bit tag: IEEE half in the fp16 build)


def _x3(ws=None):
    return _mode(ws) == 'bf16x3'


def _w5(w):
    w = np.asarray(w, dtype=np.float32)
    return w if w.ndim == 5 else w[:, :, None]  # 2D conv weight [o,i,k,k] -> [o,i,1,k,k]


def _w5d(w):
    """device variant: the fp32 master tensor (trained in place, training.Trainer) viewed as [o,i,kt,k,k]"""
    return w if w.dim() == 5 else w.unsqueeze(2)


class Executor(object):
    """Interprets one recorded net.  Prepared layers (packed weights) are cached in the workspace."""

    # consumers that read a conv / FC output only up to its real channel count (never the zero padding up to the channel stride)
    _PAD_BLIND = frozenset(['Softmax', 'Sigmoid', 'BilinearInterpolation', 'GenerateProposals', 'TimeMean', 'RpnDeltasPerFrame'])
    training = False            # (TrainExecutor: True -- gradients are taken over whole channel strides)
    _TSEL_AWARE = frozenset(['Conv', 'RoIFeatureTransform'])    # handlers that understand Blob.tsel

    def __init__(self, ws, net):
        self.ws, self.net = ws, net
        self.pending_rpn = []
        self.has_collect = any(op.type == 'CollectAndDistributeFpnRpnProposals' for op in net.ops)

    def run(self):
        self._plan_rpn_siblings()
        self._plan_keyframe_dce()
        start = self._run_cached_trunk()
        ready = getattr(self.ws, 'trunk_ready', None)
        if ready is not None:
            # the pipelined engine assembled the per-frame prefix's output itself (core/pipeline.FrameTrunkCache: gathered from per-frame
            # cache slots into a static buffer): bind it and run the rest of the net
            self.ws.trunk_ready = None
            start, live, t, N, T, C, dt = ready
            self.ws.blobs[live] = Blob(t, 'fmap', N, T, C, dt, True)
        blobs = self.ws.blobs
        for i, op in enumerate(self.net.ops):
            if i < start or i in self._skip:
                continue
            if op.type not in self._TSEL_AWARE:
                # a lazy SliceKeyFrame blob still holds all T frames of every clip: only the handlers that address frame k themselves may
                # read one (`_slice_readers_address_frames` decides per blob; this catches a reader it was never told about)
                for n in op.inputs:
                    b = blobs.get(n)
                    assert b is None or b.tsel is None, 'op %d (%s) reads the lazily sliced blob %r' % (i, op.type, n)
            getattr(self, 'op_' + op.type)(i, op)

    # ---- per-frame trunk cache (cfg.HIP.FRAME_TRUNK_CACHE) -----------------------------------------------------------
    @staticmethod
    def trunk_split(net):
        """(n_ops, blob): the leading ops of `net` that act on every frame independently (the [1,7,7] stem, spatial
        pooling, convs with time kernel 1 and their fused affine / residual / ReLU, StopGradient) and the single blob
        later ops read from them; (0, None) when the net has no such prefix."""
        n_max, produced = 0, []
        have = set()
        for op in net.ops:
            a = op.args if isinstance(op.args, dict) else {}
            ok = (op.type == 'Conv' and a.get('kernels', [2])[0] == 1 and op.inputs[0] in have | {'data'} and
                  (not a.get('residual') or a['residual'] in have) and not a.get('res_mode') == 2) or \
                 (op.type in ('MaxPool', 'StopGradient') and op.inputs[0] in have)
            if not ok:
                break
            have.update(op.outputs)
            produced.append(set(op.outputs))
            n_max += 1
        for n in range(n_max, 0, -1):       # the longest prefix with exactly one blob read by the rest of the net
            made = set().union(*produced[:n])
            live = set()
            for op in net.ops[n:]:
                a = op.args if isinstance(op.args, dict) else {}
                for b in list(op.inputs) + ([a['residual']] if a.get('residual') else []):
                    if b in made:
                        live.add(b)
            if len(live) == 1:
                return n, live.pop()
        return 0, None

    def _run_cached_trunk(self):
        """With a trunk request (ws.trunk_request = (frame ids of the clip, ids of the frames in the fed `data` blob)): run
        the per-frame prefix on the NEW frames only, keep their output per frame id (LRU), assemble the clip's blob from
        the cache and return the index of the first op that still has to run."""
        ws = self.ws
        req = getattr(ws, 'trunk_request', None)
        if req is None:
            return 0
        ws.trunk_request = None
        n, live = self.trunk_split(self.net)
        assert n > 0 and not cfg.HIP.KEYFRAME_DCE, 'frame-trunk cache: no per-frame prefix in this net (or combined with KEYFRAME_DCE)'
        ids, new_ids = req
        cache = ws.trunk_cache
        if new_ids:
            assert ws.blobs['data'].t.shape[2] == len(new_ids), (ws.blobs['data'].t.shape, len(new_ids))
            for i in range(n):
                if i not in self._skip:
                    getattr(self, 'op_' + self.net.ops[i].type)(i, self.net.ops[i])
            out = ws.blobs[live]
            assert out.N == 1 and out.t.shape[0] == len(new_ids)
            for j, fid in enumerate(new_ids):
                cache[fid] = (out.t[j:j + 1], out.C, out.dt)
                cache.move_to_end(fid)
        frames = []
        for fid in ids:
            assert fid in cache, 'frame %r is neither cached nor among the new frames' % (fid,)
            cache.move_to_end(fid)
            frames.append(cache[fid])
        t = torch.cat([f[0] for f in frames], dim=0)
        ws.blobs[live] = Blob(t, 'fmap', 1, len(ids), frames[0][1], frames[0][2], True)
        while len(cache) > max(int(cfg.HIP.FRAME_TRUNK_CACHE), len(set(ids))):
            cache.popitem(last=False)
        return n

    # ---- RPN head sibling fusion: logits + deltas 1x1 convs on the same input run as ONE conv ------------------------
    def _plan_rpn_siblings(self):
        self._skip, self._fused, self._per_frame = set(), {}, set()
        ops_ = self.net.ops
        prod = {}
        for i, op in enumerate(ops_):
            for o in op.outputs:
                prod[o] = i
        for gi, gp in enumerate(ops_):
            if gp.type != 'GenerateProposals':
                continue
            probs, deltas = gp.inputs[0], gp.inputs[1]
            si = prod.get(probs)
            if si is None or ops_[si].type != 'Sigmoid':
                continue
            li, di = prod.get(ops_[si].inputs[0]), prod.get(deltas)
            if li is None or di is None:
                continue
            ti = ri = None
            if ops_[li].type == 'TimeMean':   # tube RPN: logits averaged over T inside the proposal kernel
                ti, li = li, prod.get(ops_[li].inputs[0])
                if li is None:
                    continue
            if ops_[di].type == 'RpnDeltasPerFrame':   # (a, xywh) x T frames read in place as (a, t, xywh)
                if ti is None:
                    continue
                ri, di = di, prod.get(ops_[di].inputs[0])
                if di is None:
                    continue
            lo, do = ops_[li], ops_[di]
            if lo.type == 'Conv' and do.type == 'Conv' and lo.inputs[0] == do.inputs[0] and \
                    lo.args['kernels'] == [1, 1, 1] == do.args['kernels'] and lo.args['residual'] is None and \
                    do.args['residual'] is None:
                first = min(li, di)
                self._skip.update({li, di, si})
                if ti is not None:
                    self._skip.update({ti, ri})
                    self._per_frame.add(gi)
                self._skip.discard(first)
                self._fused[first] = (lo, do, gi)

    # ---- opt-in dead-frame elimination (cfg.HIP.KEYFRAME_DCE) -------------------------------------------------------
    def _plan_keyframe_dce(self):
        """self._keyframe[blob] = k when every reader of `blob` (in any net of this workspace) is a SliceKeyFrame at
        frame k, directly or through frame-wise ops (MaxPool with time kernel 1, FPN3D.py:155-164)."""
        self._keyframe = {}
        if not cfg.HIP.KEYFRAME_DCE:
            return
        readers = {}
        for net in self.ws.nets.values():
            for op in net.ops:
                for b in op.inputs:
                    readers.setdefault(b, []).append(op)
                res = op.args.get('residual') if isinstance(op.args, dict) else None
                if res:
                    readers.setdefault(res, []).append(op)

        def need(blob, depth=0):
            ks = set()
            rs = readers.get(blob, [])
            if not rs or depth > 4:
                return None
            for op in rs:
                if op.type == 'SliceKeyFrame':
                    ks.add(op.args['keyframe'])
                elif op.type == 'MaxPool' and op.inputs[0] == blob:
                    k = need(op.outputs[0], depth + 1)
                    if k is None:
                        return None
                    ks.add(k)
                else:
                    return None
            return ks.pop() if len(ks) == 1 else None

        for op in self.net.ops:
            if op.type in ('Conv', 'MaxPool'):
                k = need(op.outputs[0])
                if k is not None:
                    self._keyframe[op.outputs[0]] = k

    # ---- helpers ---------------------------------------------------------------------------------------------------------
    def _layer(self, key, build):
        k = (self.net.name, key)
        if k not in self.ws._layers:
            self.ws._layers[k] = build()
            # the layer's packing kernels were enqueued on THIS stream; the cache is shared by every fork of the workspace (the pipeline's
            # slots, the frame-trunk stream), whose streams would otherwise be free to launch the layer before its weights are packed.
            # Layers are built once per model: a host wait per build costs nothing in steady state.
            # (inference only: a training executor runs on one stream and rebuilds some layers every iteration)
            if not self.training and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().synchronize()
        return self.ws._layers[k]

    def _log_conv(self, name, layer, frames, H, W, oframes=None, res_mode=0):
        """(name, algorithmic flops, algorithmic HBM bytes) of one conv launch, for bench.py's roofline leg."""
        if self.ws.conv_log is not None:
            of = frames if oframes is None else oframes
            self.ws.conv_log.append((name, layer.flops(of, H, W), layer.hbm_bytes(frames, H, W, oframes, res_mode)))

    # ---- ops ---------------------------------------------------------------------------------------------------------------
    def op_Conv(self, i, op):
        ws, a = self.ws, op.args
        if i in self._fused:
            return self._rpn_head_conv(i)
        xin = ws.blobs[op.inputs[0]]
        dt = _dt(self.ws)
        if op.inputs[0] == 'data':
            return self._stem(i, op, xin)
        assert xin.kind == 'fmap', (op, xin.kind)
        if xin.t2c:
            return self._conv_over_time_channels(i, op, xin)

        def build():
            w = _w5d(ws.dev_param(a['w']))
            scale = ws.dev_param(a['scale']) if a['scale'] else None
            bias = ws.dev_param(a['shift']) if a['shift'] else (ws.dev_param(a['b']) if a['b'] else None)
            return ops.ConvLayer(w, scale, bias, stride=a['strides'], pads=a['pads'], relu=a['relu'], dtype=dt,
                                 cin_stride=xin.t.shape[3], x3=_x3(self.ws))
        layer = self._layer(i, build)
        res = ws.blobs[a['residual']].t if a['residual'] else None
        if xin.tsel is not None:    # lazy SliceKeyFrame: a kT = 1 conv computes output frame k of every clip from input frame k
            kf, tfull = xin.tsel
            assert a['kernels'][0] == 1 and res is None
            self._log_conv(op.outputs[0], layer, xin.t.shape[0], xin.t.shape[1], xin.t.shape[2], oframes=xin.N)
            y = layer(xin.t, T=tfull, out_t=(kf, 1))
            b = Blob(y, 'fmap', xin.N, 1, a['dim_out'], dt, False)
            b.count = xin.count
            ws.blobs[op.outputs[0]] = b
            return
        k = self._keyframe.get(op.outputs[0])
        if k is not None and xin.T > 1 and xin.keyframe is None and res is None:
            self._log_conv(op.outputs[0], layer, xin.t.shape[0], xin.t.shape[1], xin.t.shape[2], oframes=xin.N)
            y = layer(xin.t, T=xin.T, out_t=(k, 1), x_split=self._split_of(xin, layer))
            b = Blob(y, 'fmap', xin.N, 1, a['dim_out'], dt, False)
            b.keyframe = k
            ws.blobs[op.outputs[0]] = b
            return
        assert xin.keyframe is None or a['kernels'][0] == 1, 'temporal conv on a key-frame-only blob'
        self._log_conv(op.outputs[0], layer, xin.t.shape[0], xin.t.shape[1], xin.t.shape[2],
                       res_mode=(a['res_mode'] or 1) if res is not None else 0)
        y = layer(xin.t, T=xin.T, residual=res, res_mode=a['res_mode'], x_split=self._split_of(xin, layer),
                  want_split=layer.x3 and self._read_by_a_conv(op.outputs[0]))
        b = Blob(y, 'fmap', xin.N, xin.T, a['dim_out'], dt, xin.five_d)
        b.split = getattr(y, '_split', None)      # (bf16x3: written by this conv's epilogue for the convs that read the blob)
        b.keyframe = xin.keyframe
        b.count = xin.count   # per-RoI heads (ResNet3D.py:301-327): the live RoI count travels with the features
        ws.blobs[op.outputs[0]] = b

    def _pad_unread(self, name):
        """True when no op of any registered net reads `name` beyond its real channels (inference only): the producing conv then
        skips the zero fill of the padding channels (8 fill launches per forward on the R-18 FPN model)."""
        if self.training:
            return False
        key = ('pad_unread', self.net.name, name)
        hit = self.ws._layers.get(key)
        if hit is None:
            readers = [op for net in list(self.ws.nets.values()) + [self.net] for op in net.ops
                       if name in op.inputs or (isinstance(op.args, dict) and op.args.get('residual') == name)]
            hit = self.ws._layers[key] = all(op.type in self._PAD_BLIND for op in readers)
        return hit

    def _read_by_a_conv(self, name):
        """Does any op of a registered net read `name` through a conv-type kernel (Conv / FC / ConvTranspose input)?"""
        key = ('conv_reader', self.net.name, name)
        hit = self.ws._layers.get(key)
        if hit is None:
            hit = self.ws._layers[key] = any(o.type in ('Conv', 'FC', 'ConvTranspose') and o.inputs and o.inputs[0] == name
                                             for net in list(self.ws.nets.values()) + [self.net] for o in net.ops)
        return hit

    @staticmethod
    def _split_of(blob, layer):
        """bf16x3 layers: the blob's hi / lo split, made once per blob (res2_1_sum feeds three convs)"""
        if not layer.x3:
            return None
        if blob.split is None:
            blob.split = ops.split_bf16x2(blob.t)
        return blob.split

    def _conv_over_time_channels(self, i, op, xin):
        """1x1 conv on a blob whose T frames were moved into channels (index t*C + c): run as a KT = T conv with no
        temporal padding that writes output frame 0 only -- the transposed copy never exists."""
        ws, a, dt = self.ws, op.args, _dt(self.ws)
        assert a['kernels'] == [1, 1, 1] and a['strides'] == [1, 1] and a['residual'] is None, op
        T, C = xin.T, xin.C

        def build():
            w = ws.dev_param(a['w']).reshape(a['dim_out'], T, C, 1, 1).permute(0, 2, 1, 3, 4).contiguous()
            bias = ws.dev_param(a['b']) if a['b'] else None
            return ops.ConvLayer(w, None, bias, stride=(1, 1), pads=(0, 0, 0), relu=a['relu'], dtype=dt,
                                 cin_stride=xin.t.shape[3], x3=_x3(self.ws))
        layer = self._layer(i, build)
        self._log_conv(op.outputs[0], layer, xin.t.shape[0], xin.t.shape[1], xin.t.shape[2], oframes=xin.N)
        y = layer(xin.t, T=T, out_t=(0, 1))
        ws.blobs[op.outputs[0]] = Blob(y, 'fmap', xin.N, 1, a['dim_out'], dt, False)

    def _stem(self, i, op, xin):
        ws, a, dt = self.ws, op.args, _dt(self.ws)
        assert a['kernels'] == [1, 7, 7] and a['strides'] == [2, 2] and a['pads'] == [0, 3, 3] and a['dim_in'] == 3, \
            'the network input must feed the [1,7,7]/[1,2,2] stem conv (ResNet3D.py:258)'
        frames = None
        if xin.kind == 'u8frames':      # the uploaded uint8 frames stand for the blob (utils.blob.FrameBlob)
            frames, five_d = xin.t, xin.t.five_d
            n, t = frames.N, frames.T
            if not five_d:
                n, t = n * t, 1
        else:
            data = xin.t
            five_d = data.dim() == 5
            if not five_d:
                data = data[:, :, None]
            n, _, t, h, w = data.shape

        def build():
            scale = ws.dev_param(a['scale']) if a['scale'] else None
            bias = ws.dev_param(a['shift']) if a['shift'] else (ws.dev_param(a['b']) if a['b'] else None)
            return ops.StemConv(_w5d(ws.dev_param(a['w'])), scale, bias, dt, relu=a['relu'])
        layer = self._layer(i, build)
        # (the fused stem kernel is not a conv3d_igemm launch: it is not part of the bench's per-launch conv log)
        pool = self._stem_pool_op(i, op)
        if frames is not None and (pool is None or not cfg.HIP.get('STEM_FROM_UINT8', True)):
            data = frames.materialise()     # no fused stem + pool here: the blob after all (dat_preprocess_frames)
            data = data if data.dim() == 5 else data[:, :, None]
            frames = None
        if pool is not None:           # conv1 has one reader, pool1: one kernel, conv1 never written (cfg.HIP.FUSE_STEM_POOL)
            pi, pop = pool
            self._skip.add(pi)
            ws.blobs.pop(op.outputs[0], None)
            # (round 6: from the uint8 frames when that is what was fed -- the fp32 blob is then never written; bit-identical pool1)
            y = layer.pooled_u8(frames) if frames is not None else layer.pooled(data.float())
            ws.blobs[pop.outputs[0]] = Blob(y, 'fmap', n, t, a['dim_out'], dt, five_d)
            return
        y = layer(data.float())
        ws.blobs[op.outputs[0]] = Blob(y, 'fmap', n, t, a['dim_out'], dt, five_d)

    def _stem_pool_op(self, i, op):
        """(index, op) of the MaxPool [1,3,3] / [1,2,2] / pad 1 that is the ONLY reader of the stem's output (in this net, and in
        any other net of this workspace that does not produce the blob itself before reading it), or None."""
        if not cfg.HIP.get('FUSE_STEM_POOL', True):
            return None
        out, found = op.outputs[0], None
        no_grad = getattr(self, 'no_grad', None)
        if no_grad is not None and out not in no_grad:
            # a training executor whose stem is NOT below a StopGradient marker (RESNETS.FREEZE_AT 0): the fused kernel has no
            # backward (bwd_MaxPool would never see the gradient), so the two layers run separately
            return None
        nets = list(self.ws.nets.values())
        if not any(net is self.net for net in nets):     # (a training executor runs a net the workspace never registered)
            nets.append(self.net)
        for net in nets:
            produced = False        # a net that computes `out` itself (conv_body_net is a clone of the body with its own stem conv)
            for j, o in enumerate(net.ops):           # reads ITS OWN copy: not a reader of this net's blob
                if net is not self.net and out in o.outputs:
                    produced = True
                res = o.args.get('residual') if isinstance(o.args, dict) else None
                if out in o.inputs or res == out:
                    if net is not self.net:
                        if produced:
                            continue
                        return None
                    if found is not None or o.type != 'MaxPool':
                        return None
                    found = (j, o)
        if found is None or found[0] <= i:
            return None
        pa = found[1].args
        if (pa['k'], pa['stride'], pa['pad']) != (3, 2, 1):
            return None
        return found

    def _rpn_head_conv(self, i):
        ws, dt = self.ws, _dt(self.ws)
        lo, do, gi = self._fused[i]
        xin = ws.blobs[lo.inputs[0]]
        A = lo.args['dim_out']

        def build():
            w = torch.cat([_w5d(ws.dev_param(lo.args['w'])), _w5d(ws.dev_param(do.args['w']))], dim=0)
            if xin.t2c:   # [O, T*C, 1, 1, 1] over time-moved-to-channels -> [O, C, T, 1, 1]
                w = w.reshape(w.shape[0], xin.T, xin.C, 1, 1).permute(0, 2, 1, 3, 4).contiguous()
            b = torch.cat([ws.dev_param(lo.args['b']), ws.dev_param(do.args['b'])], dim=0)
            return ops.ConvLayer(w, None, b, stride=(1, 1), pads=(0, 0, 0), relu=False, dtype=dt,
                                 cin_stride=xin.t.shape[3], x3=_x3(self.ws))
        # one layer for all the FPN levels that share these parameters (model_builder.py: the RPN heads of the levels above the first
        # reuse its weights): in training the layer is rebuilt after every update -- once per iteration, not once per level
        layer = self._layer(('rpnhead', lo.args['w'], do.args['w'], lo.args['b'], do.args['b'], int(xin.t.shape[3]),
                             (xin.T, xin.C) if xin.t2c else None), build)
        if xin.t2c:
            self._log_conv(lo.outputs[0] + '+' + do.outputs[0], layer, xin.t.shape[0], xin.t.shape[1], xin.t.shape[2],
                           oframes=xin.N)
            y = layer(xin.t, T=xin.T, out_t=(0, 1), zero_pad=self.training)
            head = Blob(y, 'fmap', xin.N, 1, A + do.args['dim_out'], dt, False)
        else:
            self._log_conv(lo.outputs[0] + '+' + do.outputs[0], layer, xin.t.shape[0], xin.t.shape[1], xin.t.shape[2])
            y = layer(xin.t, T=xin.T, zero_pad=self.training)   # (read by the proposal / loss kernels through channel offsets only)
            head = Blob(y, 'fmap', xin.N, xin.T, A + do.args['dim_out'], dt, xin.five_d)
        ws.blobs[lo.outputs[0] + '+' + do.outputs[0]] = head
        ws.blobs['_rpnhead_for_%d' % gi] = head

    def op_MaxPool(self, i, op):
        x = self.ws.blobs[op.inputs[0]]
        y = ops.maxpool_hw(x.t, x.dt, op.args['k'], op.args['stride'], op.args['pad'])
        b = Blob(y, 'fmap', x.N, x.T, x.C, x.dt, x.five_d)
        b.keyframe = x.keyframe
        self.ws.blobs[op.outputs[0]] = b

    def op_SliceKeyFrame(self, i, op):
        x = self.ws.blobs[op.inputs[0]]
        k = op.args['keyframe']
        if x.keyframe is not None:   # producer already computed only this frame (cfg.HIP.KEYFRAME_DCE)
            assert x.keyframe == k
            self.ws.blobs[op.outputs[0]] = Blob(x.t, 'fmap', x.N, 1, x.C, x.dt, False)
            return
        f, h, w, c = x.t.shape
        if x.N > 1 and x.T > 1 and self._slice_readers_address_frames(op.outputs[0]):
            # several clips per forward: the key frames are not contiguous, and the copy (132 MB r + w for P2 at 4 clips) is not needed --
            # every reader (the shared RPN conv, RoIAlign of the box / keypoint heads) can address frame k of a T-frame map itself
            b = Blob(x.t, 'fmap', x.N, 1, x.C, x.dt, False)
            b.tsel = (k, x.T)
            self.ws.blobs[op.outputs[0]] = b
            return
        y = x.t.view(x.N, x.T, h, w, c)[:, k].contiguous() if x.N > 1 else x.t[k:k + 1]
        self.ws.blobs[op.outputs[0]] = Blob(y, 'fmap', x.N, 1, x.C, x.dt, False)

    def _slice_readers_address_frames(self, name):
        """Can the output of a SliceKeyFrame stay a view of the T-frame blob?  Yes when every op that reads it (in any registered net) is a
        conv without temporal extent / residual or a RoIFeatureTransform, in an inference executor on plain tensors (bf16 / fp32)."""
        if self.training or _x3(self.ws):
            return False
        key = ('slice_lazy', self.net.name, name)
        hit = self.ws._layers.get(key)
        if hit is None:
            ok = True
            for net in list(self.ws.nets.values()) + [self.net]:
                for j, o in enumerate(net.ops):
                    a = o.args if isinstance(o.args, dict) else {}
                    if a.get('residual') == name:
                        ok = False
                    if name not in o.inputs:
                        continue
                    if o.type == 'Conv' and a.get('kernels', [0])[0] == 1 and o.inputs[0] == name:
                        continue
                    if o.type == 'RoIFeatureTransform' and name in o.inputs[:a['n_feat']]:
                        continue
                    ok = False
            hit = self.ws._layers[key] = ok
        return hit

    def op_TimePoolAvg(self, i, op):
        x = self.ws.blobs[op.inputs[0]]
        y = ops.time_avg(x.t, x.dt, x.N, x.T)
        self.ws.blobs[op.outputs[0]] = Blob(y, 'fmap', x.N, 1, x.C, x.dt, False)

    def op_TimeToBatch(self, i, op):
        self.ws.blobs[op.outputs[0]] = self.ws.blobs[op.inputs[0]]

    def op_StopGradient(self, i, op):
        pass   # marker for the training executor (ResNet3D.py:273-274); nothing to run

    def op_TimeToChannel(self, i, op):
        x = self.ws.blobs[op.inputs[0]]
        if x.kind == 'mat':     # (R, K, T, M, M) keypoint maps (op_BatchToTime) -> (R, T*K, M, M): back to the memory order they were written in
            assert x.five_d and x.t.dim() == 5
            v = x.t.permute(0, 2, 1, 3, 4)
            assert v.is_contiguous()
            self.ws.blobs[op.outputs[0]] = Blob(v.reshape(v.shape[0], v.shape[1] * v.shape[2], v.shape[3], v.shape[4]), 'mat')
            return
        b = Blob(x.t, 'fmap', x.N, x.T, x.C, x.dt, False)
        b.t2c = True
        self.ws.blobs[op.outputs[0]] = b

    def op_SpatialMean(self, i, op):
        # [R*T,H,W,Cs] -> [R*T,1,1,Cs]; stays a feature map so the 1x1x1 output convs run on it (ResNet3D.py:318-325)
        x = self.ws.blobs[op.inputs[0]]
        f, cs = x.t.shape[0], x.t.shape[3]
        m = ops.spatial_mean(x.t, x.dt, cs)                       # fp32 [R*T, Cs]
        y = m if x.dt == ops.F32 else m.to(ops.tdtype(x.dt))
        b = Blob(y.view(f, 1, 1, cs), 'fmap', x.N, x.T, x.C, x.dt, x.five_d)
        b.count = x.count
        self.ws.blobs[op.outputs[0]] = b

    def _head_rows(self, x):
        """[R*T,1,1,Cs] head output -> fp32 [R, T, C]."""
        f, h, w, cs = x.t.shape
        assert h == 1 and w == 1, 'tube head outputs are 1x1 spatial (model_builder.py:427-473)'
        return x.t.view(x.N, x.T, cs)[:, :, :x.C].float()

    def op_TimeMean(self, i, op):
        # class scores of the T frames averaged (ReduceBackMean over T, model_builder.py:436-441)
        x = self.ws.blobs[op.inputs[0]]
        m = self._head_rows(x)
        y = ops.time_avg(m.contiguous().view(x.N * x.T, 1, 1, x.C), ops.F32, x.N, x.T).view(1, 1, x.N, x.C)
        b = Blob(y, 'rows', 1, 1, x.C, ops.F32)
        b.count = x.count
        self.ws.blobs[op.outputs[0]] = b

    def op_TubeDeltasToRows(self, i, op):
        # R,(K,4),T,1,1 -> R,(K,T,4) (model_builder.py:446-473: ExpandDims/Reshape/Transpose/ReduceBackMean x2)
        x = self.ws.blobs[op.inputs[0]]
        m = self._head_rows(x)                                    # [R, T, K*4]
        R, T, K4 = m.shape
        y = m.view(R, T, K4 // 4, 4).permute(0, 2, 1, 3).reshape(R, K4 * T).contiguous()
        b = Blob(y, 'mat')
        b.count = x.count
        self.ws.blobs[op.outputs[0]] = b

    def op_Alias(self, i, op):
        self.ws.blobs[op.outputs[0]] = self.ws.blobs[op.inputs[0]]

    def op_Sigmoid(self, i, op):
        x = self.ws.blobs[op.inputs[0]]
        b = Blob(x.t, x.kind, x.N, x.T, x.C, x.dt, x.five_d)
        b.sigmoid_of = op.inputs[0]
        self.ws.blobs[op.outputs[0]] = b

    def op_GenerateProposals(self, i, op):
        ws = self.ws
        head = ws.blobs.get('_rpnhead_for_%d' % i)
        assert head is not None, 'GenerateProposals expects the fused logits+deltas head conv (op %d)' % i
        anchors = op.args['anchors']
        A, T = anchors.shape[0], anchors.shape[1] // 4
        f, h, w, cs = head.t.shape
        key = ('anchors', i)
        an = self._layer(key, lambda: ops.torch.from_numpy(anchors.astype(np.float32)).to(ws.device))
        per_frame = i in self._per_frame
        assert (head.T == T) if per_frame else (head.T == 1), \
            'tube RPN head with %d frames for %d-frame anchors (per_frame %s)' % (head.T, T, per_frame)
        spec = ops.RpnLevelSpec(head.t, h, w, A, T, 1. / op.args['spatial_scale'], cs, 0, A, 0, an, apply_sigmoid=True,
                                per_frame=per_frame)
        self._rpn_images, self._rpn_frame_stride = head.N, head.T      # image i of the batch: frames [i * head.T, (i + 1) * head.T)
        assert f == head.N * head.T, (f, head.N, head.T)
        self.pending_rpn.append((spec, op))
        if not self.has_collect:
            self._run_rpn(single=True)

    def _run_rpn(self, single=False):
        ws = self.ws
        key = 'TRAIN' if self.net._helper is not None and self.net._helper.train else 'TEST'
        im_info = ws.blobs['im_info']
        info = im_info.host if im_info.host is not None else im_info.t.cpu().numpy()
        specs = [s for s, _ in self.pending_rpn]
        # several images per forward (the reference's op loops over them, generate_proposals.py:133-147; its inference is batch 1,
        # core/test.py:212-214): the image is a grid dimension of the same kernels, every image keeps what it keeps alone
        ni = int(info.shape[0])
        assert ni == self._rpn_images, 'im_info has %d rows for %d images in the RPN head blobs' % (ni, self._rpn_images)
        assert ni == 1 or key == 'TEST', 'several images per forward: inference only (TRAIN.IMS_PER_BATCH 1 per GPU)'
        rois, probs, counts = ops.rpn_proposals(specs, _dt(self.ws), info, cfg[key].RPN_PRE_NMS_TOP_N,
                                                cfg[key].RPN_POST_NMS_TOP_N, cfg[key].RPN_NMS_THRESH,
                                                cfg[key].RPN_MIN_SIZE, n_images=ni, frame_stride=self._rpn_frame_stride)
        for li, (_, op) in enumerate(self.pending_rpn):
            if ni == 1:
                ws.blobs[op.outputs[0]] = Blob(rois[li], 'rois', count=counts[li:li + 1])
                if len(op.outputs) > 1:
                    ws.blobs[op.outputs[1]] = Blob(probs[li].view(-1, 1), 'mat', count=counts[li:li + 1])
            else:       # strided views [n_images, post_nms, ...] (no copy; only read when somebody fetches a per-level blob)
                ws.blobs[op.outputs[0]] = Blob(rois[:, li], 'rois', count=counts[:, li])
                if len(op.outputs) > 1:
                    ws.blobs[op.outputs[1]] = Blob(probs[:, li].unsqueeze(-1), 'mat', count=counts[:, li])
        self._rpn_out = (rois, probs, counts)
        self.pending_rpn = []

    def op_CollectAndDistributeFpnRpnProposals(self, i, op):
        ws = self.ws
        self._run_rpn()
        rois, probs, counts = self._rpn_out
        key = 'TRAIN' if self.net._helper is not None and self.net._helper.train else 'TEST'
        out, n_out = ops.collect_rois(rois, probs, counts, cfg[key].RPN_POST_NMS_TOP_N)
        ws.blobs['rois'] = Blob(out, 'rois', count=n_out)
        # rois_fpn<l> / rois_idx_restore_int32 are derived lazily on the host if somebody fetches them; the device
        # RoIAlign assigns levels in-kernel (no Concat + BatchPermutation, detector.py:283-296)
        for name in op.outputs[1:]:
            ws.blobs.pop(name, None)

    def op_RoIFeatureTransform(self, i, op):
        ws, a = self.ws, op.args
        feats = [ws.blobs[n] for n in op.inputs[:a['n_feat']]]
        rois = ws.blobs[op.inputs[-1]]
        rt = rois.t
        if rt.dim() != 2:
            rt = rt.view(-1, rt.shape[-1])
        R, cols = rt.shape
        Tr = (cols - 1) // 4
        f0 = feats[0]
        assert Tr == 1 or Tr == f0.T, 'tube rois of %d frames on features with T=%d' % (Tr, f0.T)
        t_feat, t0 = f0.T, 0
        if f0.tsel is not None:     # lazy SliceKeyFrame: the maps still hold all T frames; RoIAlign reads frame k of the roi's clip
            assert all(f.tsel == f0.tsel for f in feats) and Tr == 1
            t0, t_feat = f0.tsel
        else:
            assert all(f.tsel is None for f in feats)
        y = ops.roi_align([f.t for f in feats], a['scales'], f0.dt, rt.float().contiguous(), T=t_feat, Tr=Tr, t0=t0,
                          pooled=a['resolution'], sampling=a['sampling_ratio'], k_min=cfg.FPN.ROI_MIN_LEVEL,
                          canon_scale=float(cfg.FPN.ROI_CANONICAL_SCALE), canon_level=cfg.FPN.ROI_CANONICAL_LEVEL)
        b = Blob(y, 'fmap', R, Tr, f0.C, f0.dt, Tr > 1)
        b.count = rois.count
        ws.blobs[op.outputs[0]] = b

    def op_FC(self, i, op):
        ws, a, dt = self.ws, op.args, _dt(self.ws)
        x = ws.blobs[op.inputs[0]]
        if x.kind == 'fmap':
            f, p, p2, cs = x.t.shape
            assert cs == x.C, 'FC over a channel-padded RoI feature is not supported'
            # a tube RoI's T frames are contiguous: one row of T*p*p*C inputs per RoI (head_builder.py:29-33)
            xin = x.t.view(1, 1, f // x.T, x.T * p * p2 * cs)
            perm = (x.C, x.T, p, p2)
        else:
            xin, perm = x.t, None

        def build():
            w = ws.dev_param(a['w'])
            if perm is not None:  # reference flattens NC[T]HW (c, t, h, w); our RoI features are (t, h, w, c)
                c, tt, hh, ww = perm
                w = w.reshape(w.shape[0], c, tt, hh, ww).permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1)
            return ops.ConvLayer(w.reshape(w.shape[0], -1, 1, 1, 1).contiguous(), None, ws.dev_param(a['b']), stride=(1, 1),
                                 pads=(0, 0, 0), relu=a['relu'], dtype=dt, cin_stride=xin.shape[3], x3=_x3(self.ws))
        layer = self._layer(i, build)
        self._log_conv(op.outputs[0], layer, 1, 1, xin.shape[2])
        y = layer(xin, T=1, zero_pad=not self._pad_unread(op.outputs[0]))
        b = Blob(y, 'rows', 1, 1, a['dim_out'], dt)
        b.count = x.count
        ws.blobs[op.outputs[0]] = b

    def _rows_to_mat(self, b):
        r, cs = b.t.shape[2], b.t.shape[3]
        return ops.to_ncdhw(b.t.view(r, 1, 1, cs), b.dt, r, b.C, 1).view(r, b.C)

    def op_Softmax(self, i, op):
        x = self.ws.blobs[op.inputs[0]]
        m = self._rows_to_mat(x)
        b = Blob(ops.softmax_rows(m, x.C), 'mat')
        b.count = x.count
        self.ws.blobs[op.outputs[0]] = b

    def op_ConvTranspose(self, i, op):
        ws, a, dt = self.ws, op.args, _dt(self.ws)
        x = ws.blobs[op.inputs[0]]
        if x.t2c:
            return self._deconv_over_time_channels(i, op, x)
        assert a.get('group', 1) == 1, 'a grouped ConvTranspose reads a time -> channel blob (model_builder.py:765-767)'

        def build():
            w3 = ops.deconv_k4s2_as_conv3x3(ws.dev_param(a['w']))
            bias = ws.dev_param(a['b']).repeat(4)
            return ops.ConvLayer(w3, None, bias, stride=(1, 1), pads=(0, 1, 1), relu=False, dtype=dt,
                                 cin_stride=x.t.shape[3], x3=_x3(self.ws))
        layer = self._layer(i, build)
        if ws.conv_log is not None:   # algorithmic flops of the deconv itself: 16 taps / 4 outputs per input position
            ws.conv_log.append((op.outputs[0], 2.0 * a['dim_in'] * a['dim_out'] * 16 * x.t.shape[0] * x.t.shape[1] * x.t.shape[2],
                                layer.hbm_bytes(x.t.shape[0], x.t.shape[1], x.t.shape[2])))
        y = layer(x.t, T=1, zero_pad=not self._pad_unread(op.outputs[0]))
        b = Blob(y, 'fmap', x.N, x.T, 4 * a['dim_out'], dt, x.five_d)
        ws.blobs[op.outputs[0]] = b

    @staticmethod
    def deconv_dense_filter(w, T, group):
        """The [T*C, T*K, 4, 4] filter of the deconv over time-moved-to-channels: `w` itself when it already has that shape (group
        dropped, cfg.HIP.DECONV_GROUP_IGNORED), else the block-diagonal expansion of the grouped filter [T*C, K, 4, 4] -- block t
        (input channels t*C.., output maps t*K..) = w[t*C:(t+1)*C]."""
        tc, kk = int(w.shape[0]), int(w.shape[1])
        if group == 1:
            return w
        assert group == T and tc % T == 0, (tuple(w.shape), T, group)
        c = tc // T
        dense = torch.zeros((tc, T * kk) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
        for t in range(T):
            dense[t * c:(t + 1) * c, t * kk:(t + 1) * kk] = w[t * c:(t + 1) * c]
        return dense

    def _deconv_group(self, a, w):
        """The group count the op EXECUTES with: the recorded one when the filter is in the grouped layout, 1 when it is the full brew blob."""
        g = int(a.get('group', 1))
        return g if (g > 1 and int(w.shape[1]) * g == a['dim_out']) else 1

    def _deconv_over_time_channels(self, i, op, x):
        """ConvTranspose k4 s2 on a blob whose T frames were moved into channels (index t*C + c; model_builder.py:765-767, :848-856 --
        the reference default KRCNN.NO_3D_DECONV_TIME_TO_CH False): the sub-pixel 3 x 3 conv with KT = T temporal taps, no temporal
        padding, output frame 0 only -- ONE launch per forward, the transposed copy never exists.  group = T (one [C, K, 4, 4] block per
        frame) runs as the block-diagonal case of the same launch (T = 3: 0.3 GFLOP per 100 rois either way).  Output: one frame per
        roi with 4*T*K sub-pixel channels, map index t*K + k."""
        ws, a, dt = self.ws, op.args, _dt(self.ws)
        T, C = x.T, x.C
        assert a['dim_in'] == T * C and a['dim_out'] % T == 0, (a['dim_in'], a['dim_out'], T, C)

        def build():
            w = ws.dev_param(a['w'])
            dense = self.deconv_dense_filter(w, T, self._deconv_group(a, w))        # [T*C, T*K, 4, 4]
            w3 = ops.deconv_k4s2_as_conv3x3(dense)                                    # [4*T*K, T*C, 1, 3, 3]
            w3 = w3.reshape(w3.shape[0], T, C, 3, 3).permute(0, 2, 1, 3, 4).contiguous()
            bias = ws.dev_param(a['b']).repeat(4)
            return ops.ConvLayer(w3, None, bias, stride=(1, 1), pads=(0, 1, 1), relu=False, dtype=dt,
                                 cin_stride=x.t.shape[3], x3=_x3(self.ws))
        layer = self._layer(i, build)
        if ws.conv_log is not None:
            g = self._deconv_group(a, ws.dev_param(a['w']))
            ws.conv_log.append((op.outputs[0], 2.0 * a['dim_in'] * a['dim_out'] / g * 16 * x.N * x.t.shape[1] * x.t.shape[2],
                                layer.hbm_bytes(x.t.shape[0], x.t.shape[1], x.t.shape[2], oframes=x.N)))
        y = layer(x.t, T=T, out_t=(0, 1), zero_pad=not self._pad_unread(op.outputs[0]))
        b = Blob(y, 'fmap', x.N, 1, 4 * a['dim_out'], dt, False)
        b.count = x.count
        ws.blobs[op.outputs[0]] = b

    def op_BilinearInterpolation(self, i, op):
        x = self.ws.blobs[op.inputs[0]]
        K = op.args['dim']
        out = ops.kps_finalize(x.t, x.dt, x.N, x.T, K, op.args['up_scale'])        # (R, T*K, M, M), map index t*K + k
        # a 3D head whose frames sit in the batch axis (model_builder.py:760-764): the blob the reference has here is (R*T, K, M, M) -- the
        # same memory; BatchToTime / TimeToChannel (:864-868) are views that end at the (R, T*K, M, M) the kernel wrote.  (T = 1: the same)
        self.ws.blobs[op.outputs[0]] = Blob(out.view(x.N * x.T, K, out.shape[2], out.shape[3]), 'mat', x.N, x.T, K)

    def op_BatchToTime(self, i, op):
        """detector.py:513-534 on the up-sampled keypoint maps: (R*T, K, M, M) -> (R, K, T, M, M), a strided view (FetchBlob copies it out
        in the reference's order; the TimeToChannel that follows undoes the transpose)."""
        x = self.ws.blobs[op.inputs[0]]
        assert x.kind == 'mat' and x.t.dim() == 4 and x.T >= 1 and x.t.shape[0] == x.N * x.T, 'BatchToTime: keypoint maps of a 3D head only'
        v = x.t.view((x.N, x.T) + tuple(x.t.shape[1:])).permute(0, 2, 1, 3, 4)
        b = Blob(v, 'mat', x.N, x.T, x.C)
        b.five_d = True
        self.ws.blobs[op.outputs[0]] = b

    def __getattr__(self, name):
        if name.startswith('op_'):
            raise NotImplementedError('executor has no handler for recorded op type %s' % name[3:])
        raise AttributeError(name)


# ---- module-level API (drop-in for `from caffe2.python import workspace`) --------------------------------------------------
_GLOBAL = None


def GlobalWorkspace():
    global _GLOBAL
    if _GLOBAL is None:
        _GLOBAL = Workspace(torch.cuda.current_device())
    return _GLOBAL


def ResetWorkspace():
    global _GLOBAL
    _GLOBAL = None


def FeedBlob(name, arr):
    GlobalWorkspace().FeedBlob(name, arr)


def FetchBlob(name):
    return GlobalWorkspace().FetchBlob(name)


def HasBlob(name):
    return GlobalWorkspace().HasBlob(name)


def CreateNet(net):
    return GlobalWorkspace().CreateNet(net)


def RunNet(name):
    GlobalWorkspace().RunNet(name)
