"""Training executor: forward with losses, backward over the recorded graph, momentum-SGD update and
the data-parallel gradient exchange.

Reference: `model_builder.build_data_parallel_model` (:908-952: AddGradientOperators + NCCLAllreduce/muji) and
`add_parameter_update_ops` (:954-985), losses :481-494 / :612-636 / :873-889 and FPN.py:282-321, run by
tools/train_net.py:120-170.  Here one process drives one GPU: the backward pass walks the op list in reverse and launches
the training kernels of `csrc/train_ops.hip` / `csrc/losses.hip` (conv weight gradient GEMMs, data gradient through the
forward MFMA kernel, fused ReLU/bias backward, RoIAlign / FPN top-down / keypoint-tail backward, loss value + gradient),
gradients are summed across ranks with ONE bucketed `torch.distributed.all_reduce` per bucket (RCCL over xGMI; the only
collective on this path, losses are pre-divided by NUM_GPUS exactly like the reference) and every rank applies the
identical local update.

Gradient semantics follow the reference's ops: AffineChannelNd has no scale/bias gradient
(lib/ops/affine_channel_nd_op.cc:29-37 'TODO'), StopGradient freezes conv1/res2 (ResNet3D.py:273-274), rois are
constants (no gradient through GenerateProposals), biases get 2x gradient and no weight decay (:968-974).
"""
import logging
import os

import numpy as np
import torch

from detectandtrack_amd.core.config import cfg
from detectandtrack_amd.ops import hip_ops as ops
from detectandtrack_amd.utils import lr_policy
from detectandtrack_amd.workspace import Executor, _count

logger = logging.getLogger(__name__)


class TrainExecutor(Executor):
    """Forward (inherits the inference handlers) + loss ops + backward + update for one net on one workspace."""
    training = True

    def __init__(self, ws, net, arena=None, gt_arena=None):
        super(TrainExecutor, self).__init__(ws, net)
        self.grads = {}          # blob name -> list of gradient tensors (blob layout; activation dtype or fp32)
        self.param_grads = {}    # param name -> fp32 CUDA tensor (reference blob layout)
        # Trainer's gradient arena: param name -> a zeroed fp32 view of ONE flat buffer (the all-reduce and the SGD update run on
        # the flat buffer; weight-gradient kernels write into the views directly).  None: gradients are separate tensors.
        self.arena = arena
        # Deferred weight-gradient finish (cfg.HIP.DEFER_WGRAD_FINISH): weight name -> a ZEROED fp32 accumulator of the weight's size
        # (views of one flat buffer the Trainer zeroes per iteration).  The conv weight-gradient kernels add into it in their own
        # [tap][Cout][Cin] order and ONE batched launch at the end of backward() writes the gradients into the arena -- instead of a
        # memset + a transposing finish launch around every layer's kernel (46 + 46 launches of an R-18 iteration).
        self.gt_arena = gt_arena if arena is not None else None
        self._deferred = {}      # weight name -> the fused AffineChannelNd scale (tensor or None) its finish multiplies by
        self._pw_pending = []    # (weight name, ConvGrad.weight_acc_job) of pointwise layers waiting for their grouped launch
        self._pw_batch = int(cfg.HIP.get('WGRAD_PW_BATCH', 16)) if gt_arena is not None and arena is not None else 0
        self._sealed = set()     # parameters whose gradient bucket is already on its way to the other ranks (Trainer, overlap)
        self.accumulate_all = False   # the arena already holds gradients of an earlier clip (Trainer.step(zero_grad=False)): add, never overwrite
        self._iter_cache = {}           # per backward pass: objects several ops build from the same weights (see backward)
        self._last_masked = False       # set by _take_grad: the gradient just taken already carries its producer's ReLU mask
        self._ncontrib = {}             # blob name -> gradient contributions received so far (entries of self.grads may be sums of several)
        self._trainable_set = None
        self._readers = {}
        for o in net.ops:
            res = o.args.get('residual') if isinstance(o.args, dict) else None
            for b in set(list(o.inputs) + ([res] if res else [])):      # (a fused Sum lists its residual among the inputs AND in its args: one reader)
                self._readers[b] = self._readers.get(b, 0) + 1
        self.losses = {}         # loss blob name -> fp32 CUDA scalar tensor
        self.metrics = {}
        self._cg = {}
        # blobs at or below the StopGradient marker (frozen trunk): no gradient is computed for them or their producers
        self.no_grad = no_grad_blobs(net)

    # ---- forward additions ---------------------------------------------------------------------------------------------
    def op_StopGradient(self, i, op):
        pass

    def _head_blob_of(self, logits_name):
        """fused logits+deltas head conv feeding `logits_name` (directly, or through the tube RPN's TimeMean)"""
        prod = self.net.producer(logits_name)
        src = prod.inputs[0] if (prod is not None and prod.type == 'TimeMean') else logits_name
        for first, (lo, do, gi) in self._fused.items():
            if lo.outputs[0] == src:
                return lo.outputs[0] + '+' + do.outputs[0], lo, do, gi
        raise KeyError(logits_name)

    def _scalar_pool(self, n):
        """n zeroed 4-byte words out of ONE buffer zeroed once per forward (hipMemsetAsync): the loss / metric accumulators of an
        iteration -- 13 losses over ~10 ops -- used to be one fill launch each."""
        pool = getattr(self, '_pool', None)
        if pool is None or self._pool_used + n > pool.numel():
            pool = self._pool = torch.empty(256, dtype=torch.float32, device=self.ws.device)
            ops.ctx().call('dat_fill_zero', ops._stream(), ops._ptr(pool), ops.C.c_size_t(pool.numel() * 4))
            self._pool_used = 0
        t = pool[self._pool_used:self._pool_used + n]
        self._pool_used += n
        return t

    def _loss_buf(self, names):
        t = self._scalar_pool(len(names))
        for j, n in enumerate(names):
            self.losses[n] = t[j:j + 1]
        return t

    def op_RpnLoss(self, i, op):
        ws, a = self.ws, op.args
        head_name, lo, do, gi = self._head_blob_of(op.inputs[0])
        head = ws.blobs[head_name]
        A = lo.args['dim_out']
        per_frame = gi in self._per_frame
        T = head.T if per_frame else do.args['dim_out'] // (4 * A)
        labels, tgt, w_in, w_out = [ws.blobs[n] for n in op.inputs[2:6]]
        cls_mult = a['cls_scale']
        if a['normalize']:
            f, h, w, _ = head.t.shape
            if hasattr(labels.host, 'count_in_window'):     # roi_data.loader: the sampled anchors are known sparsely
                cls_mult /= max(1.0, float(labels.host.count_in_window(h, w)))
            else:
                lab = labels.host if labels.host is not None else labels.t.cpu().numpy()
                cls_mult /= max(1.0, float((lab[:, :, :h, :w] >= 0).sum()))
        n_batch = head.N
        loss2 = self._loss_buf(op.outputs)
        dhead = ops.rpn_loss(head.t, head.dt, A, 0, A, labels.t, tgt.t, w_in.t, w_out.t, cls_mult, a['beta'],
                             a['bbox_scale'] / n_batch, loss2, T=T, per_frame=per_frame)
        self._add_grad(head_name, dhead)

    def op_SoftmaxLoss(self, i, op):
        ws, a = self.ws, op.args
        x = ws.blobs[op.inputs[0]]            # rows [1,1,R,cs]
        labels = ws.blobs[op.inputs[1]]
        R = x.t.shape[2]
        loss = self._loss_buf([op.outputs[1]])
        correct = self._scalar_pool(1).view(torch.int32)
        d = ops.softmax_ce_rows(x.t, x.dt, x.C, labels.t, None, a['scale'] / max(R, 1), loss, correct)
        self.metrics[op.outputs[2]] = (correct, R)
        self._add_grad(op.inputs[0], d)

    def op_SmoothL1Loss(self, i, op):
        ws, a = self.ws, op.args
        x = ws.blobs[op.inputs[0]]
        tgt, w_in, w_out = [ws.blobs[n] for n in op.inputs[1:4]]
        loss = self._loss_buf(op.outputs)
        if x.kind == 'mat':      # tube box deltas regrouped to rows (TubeDeltasToRows): fp32 [R, K*T*4]
            R, D = x.t.shape
            d = ops.smooth_l1_rows(x.t.contiguous(), ops.F32, D, tgt.t, w_in.t, w_out.t, a['beta'], a['scale'] / max(R, 1), loss)
        else:
            R = x.t.shape[2]
            d = ops.smooth_l1_rows(x.t, x.dt, x.C, tgt.t, w_in.t, w_out.t, a['beta'], a['scale'] / max(R, 1), loss)
        self._add_grad(op.inputs[0], d)

    def op_KeypointLoss(self, i, op):
        ws, a = self.ws, op.args
        x = ws.blobs[op.inputs[0]]            # 'mat' fp32 [R, K, M, M]
        loc, wts = ws.blobs[op.inputs[1]], ws.blobs[op.inputs[2]]
        R, K, M, _ = x.t.shape
        if hasattr(wts.host, 'weight_sum'):         # roi_data.device_sampler: the sum came back with the roi counts
            norm = float(wts.host.weight_sum)
        else:
            w_host = wts.host if wts.host is not None else wts.t.cpu().numpy()
            norm = float(np.sum(w_host))
        loss = self._loss_buf([op.outputs[1]])
        logits = x.t.view(R * K, M * M)
        mult = a['scale'] / norm if norm > 0 else 0.0
        d = ops.softmax_ce_rows(logits, ops.F32, M * M, loc.t.view(-1), wts.t.view(-1), mult, loss)
        self._add_grad(op.inputs[0], d.view(R, K, M, M))

    def op_CollectAndDistributeFpnRpnProposals(self, i, op):
        if not op.args.get('train'):
            return super(TrainExecutor, self).op_CollectAndDistributeFpnRpnProposals(i, op)
        ws = self.ws
        self._run_rpn()
        rois, probs, counts = self._rpn_out
        out, n_out = ops.collect_rois(rois, probs, counts, cfg.TRAIN.RPN_POST_NMS_TOP_N)
        im_info = ws.blobs['im_info']
        info = im_info.host if im_info.host is not None else im_info.t.cpu().numpy()
        sampler = getattr(ws, 'train_sampler', None)
        assert sampler is not None, 'training needs ws.train_sampler(rois, im_info) -> dict of sampled blobs ' \
                                    '(roi_data.fast_rcnn.add_fast_rcnn_blobs on the roidb entry of this clip)'
        if getattr(sampler, 'on_device', False):    # roi_data.device_sampler: the proposals never leave the GPU, the row counts come back
            self._feed_device_samples(op, sampler(out, n_out, info))
            return
        rois_np = out[:int(n_out.item())].cpu().numpy()
        blobs = sampler(rois_np, info)
        for name in op.outputs:
            if name in blobs:
                ws.FeedBlob(name, np.ascontiguousarray(blobs[name]))

    def _feed_device_samples(self, op, blobs):
        from detectandtrack_amd.roi_data.device_sampler import WeightSum
        for name in op.outputs:
            if name in blobs:
                self.ws.FeedBlob(name, blobs[name])
        if 'keypoint_weights' in blobs and 'keypoint_weights' in self.ws.blobs:
            self.ws.blobs['keypoint_weights'].host = WeightSum(blobs['keypoint_weights_sum'])

    def op_GenerateProposalLabels(self, i, op):
        """ops/generate_proposal_labels.py:23-37: sample labelled training rois from the RPN proposals of this clip."""
        ws = self.ws
        r = ws.blobs[op.inputs[0]]
        im_info = ws.blobs['im_info']
        info = im_info.host if im_info.host is not None else im_info.t.cpu().numpy()
        assert ws.train_sampler is not None, 'training needs ws.train_sampler(rois, im_info) -> dict of sampled blobs'
        if getattr(ws.train_sampler, 'on_device', False) and r.count is not None:
            self._feed_device_samples(op, ws.train_sampler(r.t if r.t.dim() == 2 else r.t.view(-1, r.t.shape[-1]), r.count, info))
            return
        rois_np = r.t[:_count(r)].cpu().numpy() if r.count is not None else r.t.cpu().numpy()
        blobs = ws.train_sampler(rois_np, info)
        for name in op.outputs:
            if name in blobs:
                ws.FeedBlob(name, np.ascontiguousarray(blobs[name]))

    # ---- gradient bookkeeping ---------------------------------------------------------------------------------------------
    # A gradient of a T-frame feature map may cover only a WINDOW of frames [lo, lo + n) (everything else is exactly zero):
    # with BODY_HEAD_LINK 'slice-center' the heads read one frame, so the FPN post-hoc convs (60 % of the forward FLOPs)
    # receive a 1-frame gradient and hand a 3-frame gradient down; each kT = 3 conv widens the window by its temporal reach.
    def _add_grad(self, name, t, lo=0, masked=False):
        """masked: the ReLU backward of `name`'s producer is already applied to `t` (fused into the data-gradient conv that made it).
        The flag travels WITH the entry (tensor, first frame, masked) -- not in a side table keyed by id(tensor)."""
        self.grads.setdefault(name, []).append((t, lo, bool(masked)))
        self._ncontrib[name] = self._ncontrib.get(name, 0) + 1

    def _take_grad(self, name, dtype):
        """-> (dy, lo) with dy in the activation dtype covering frames [lo, lo + dy.shape[0]), or (None, 0).  `self._last_masked`
        says whether the returned gradient already carries the ReLU mask of `name`'s producer (a single, masked contribution)."""
        lst = self.grads.pop(name, None)
        self._ncontrib.pop(name, None)
        self._last_masked = False
        if not lst:
            return None, 0
        tdt = ops.tdtype(dtype)
        self._last_masked = len(lst) == 1 and lst[0][2]
        assert not any(m for _, _, m in lst) or len(lst) == 1, 'a masked gradient must be the only contribution to %r' % name
        lo = min(l for _, l, _ in lst)
        hi = max(l + t.shape[0] for t, l, _ in lst)
        if len(lst) == 1:
            t, l, _ = lst[0]
            return (t if t.dtype == tdt else t.to(tdt)), l
        # the sum starts as ONE out-of-place add of two full-window contributions (no clone + add pair).  (Adding the fp32 accumulators of
        # the RoIAlign backward without the .to() pass -- a mixed-dtype in-place add -- measured slower: ATen's casting kernel is not vectorised.)
        full = [e for e in lst if e[0].shape[0] == hi - lo and not getattr(e[0], '_roi_acc', False)]
        first = next((e for e in full if e[0].dtype == tdt), None)
        acc = None
        if first is not None:
            second = next((e for e in full if e is not first and e[0].shape == first[0].shape), None)
            if second is not None:
                acc = torch.add(first[0], second[0]) if second[0].dtype == tdt else torch.add(first[0], second[0]).to(tdt)
                lst = [e for e in lst if e is not first and e is not second]
            else:
                acc = first[0].clone()
                lst = [e for e in lst if e is not first]
        if acc is None:
            acc = torch.zeros((hi - lo,) + tuple(lst[0][0].shape[1:]), dtype=tdt, device=lst[0][0].device)
        for t, l, _ in lst:
            acc[l - lo:l - lo + t.shape[0]] += t if t.dtype == tdt else t.to(tdt)
        return acc, lo

    def _pgrad(self, name, t):
        """Add a parameter-gradient contribution `t` (a fresh tensor the caller does not reuse)."""
        if name in self._sealed:
            raise RuntimeError('gradient contribution to %r after its bucket was handed to the all-reduce (static completion index '
                               'of the parameter is wrong)' % name)
        if self.arena is not None and name in self.arena:
            view = self.arena[name]
            if t.data_ptr() != view.data_ptr():      # not produced in place (see _pgrad_out)
                view += t.reshape(view.shape)
            self.param_grads[name] = view
            return
        t = t.reshape(self.ws.params[name].shape) if name in self.ws.params else t
        if name in self.param_grads:
            self.param_grads[name] += t
        else:
            self.param_grads[name] = t

    def _pgrad_out(self, name):
        """Where a kernel may WRITE (overwrite) the gradient of `name` directly: its arena view on the first contribution of the
        step, else None (the kernel then returns a temporary that _pgrad adds)."""
        if self.arena is None or name not in self.arena or name in self.param_grads or self.accumulate_all:
            return None
        return self.arena[name]

    def _conv_grad(self, i, build):
        """ConvGrad of op i, cached across iterations in the workspace (the Trainer re-packs it after every update)."""
        key = (self.net.name, ('grad', i))
        if key not in self.ws._layers:
            self.ws._layers[key] = build()
        return self.ws._layers[key]

    def _master(self, name):
        """fp32 device master copy of a parameter (reference blob layout)."""
        return self.ws.dev_param(name)

    def _trainable(self, pname):
        if self._trainable_set is None:        # (was rebuilt on every call: 0.8 ms of host time per iteration)
            self._trainable_set = set(self.net._helper.TrainableParams())
        return pname is not None and pname in self._trainable_set

    # ---- backward -----------------------------------------------------------------------------------------------------------
    def backward(self, on_op_done=None):
        """on_op_done(i): called after the backward of op i (ops run from the last to the first) -- the Trainer's hook for finishing
        and exchanging gradient buckets while the backward of the earlier layers continues."""
        self._iter_cache = {}        # objects built from this iteration's weights that several ops share (the RPN heads of the FPN levels)
        for i in range(len(self.net.ops) - 1, -1, -1):
            op = self.net.ops[i]
            skip = (i in self._skip and i not in self._fused) or (op.outputs and all(o in self.no_grad for o in op.outputs))
            if not skip:
                h = getattr(self, 'bwd_' + op.type, None)
                if h is not None:
                    h(i, op)
            if on_op_done is not None:
                on_op_done(i)
        self._finish_deferred()
        self.grads.clear()
        self._ncontrib.clear()

    def _flush_pw(self):
        """Run the queued pointwise weight gradients (one grouped launch per tile class) and drop the references that kept their operands alive."""
        if self._pw_pending:
            ops.wgrad_acc_batch([j for _, j in self._pw_pending])
            self._pw_pending = []

    def _finish_deferred(self, only=None):
        """Turn the deferred weight-gradient accumulators into gradients (one batched launch).  only: restrict to these parameter
        names (one gradient bucket); the others stay deferred."""
        if self._pw_pending and (only is None or any(n in only for n, _ in self._pw_pending)):
            self._flush_pw()        # (the accumulators must be complete before they are read)
        names = sorted(n for n in self._deferred if only is None or n in only)
        if not names:
            return
        ent = [(self.gt_arena[n], self._deferred[n], self.arena[n], n in self.param_grads or self.accumulate_all) for n in names]
        key = tuple((n, bool(acc), gt.data_ptr(), sc.data_ptr() if sc is not None else 0, dw.data_ptr())
                    for n, (gt, sc, dw, acc) in zip(names, ent))
        cache = self.ws.__dict__.setdefault('_wfinish_cache', {})
        if key not in cache:
            if len(cache) >= 64:                # (tables of an earlier configuration: they point at buffers that may be gone)
                cache.clear()
            cache[key] = ops.WeightFinishBatch(ent)
        cache[key].run()
        for n in names:
            self.param_grads[n] = self.arena[n]
            del self._deferred[n]

    def seal(self, names):
        """The gradients of `names` are final (their bucket is being exchanged): a later contribution is a bug, not a silent loss."""
        self._sealed.update(names)

    def bwd_Conv(self, i, op):
        ws, a = self.ws, op.args
        if i in self._fused:
            return self._bwd_rpn_head(i)
        out = op.outputs[0]
        y = ws.blobs[out]
        dy, lo = self._take_grad(out, y.dt)
        if dy is None:
            return
        xin = ws.blobs[op.inputs[0]]
        assert not xin.t2c and y.keyframe is None, 'training with time->channel heads / key-frame DCE is not supported'
        n = dy.shape[0]
        full = (lo == 0 and n == y.t.shape[0])
        assert full or xin.N == 1, 'frame-window gradients assume one clip per forward (TRAIN.IMS_PER_BATCH 1)'
        cout = a['dim_out']
        train_b = a['b'] if (a['b'] and self._trainable(a['b'])) else None
        dbias = None
        if train_b:     # the kernel ACCUMULATES the bias reduction: straight into the (zeroed) arena view when there is one
            dbias = self.arena[train_b] if (self.arena is not None and train_b in self.arena) else \
                torch.zeros(cout, dtype=torch.float32, device=ws.device)
        # (the mask may already be in dy: the data-gradient conv of this blob's only reader applied it in its epilogue)
        need_relu = bool(a['relu']) and not self._last_masked
        if need_relu or dbias is not None:
            g = ops.relu_bias_bwd(dy, y.t[lo:lo + n], y.dt, cout, relu=need_relu, dbias=dbias)
        else:
            g = dy          # no ReLU to mask by, no trainable bias to reduce into (the shortcut convs: AffineChannelNd has no gradient)
            # g ALIASES dy here (it may also be queued as the residual's gradient below): everything downstream reads whole channel
            # strides, so the padding channels [cout, stride) must be zero -- they are when dy came out of a conv epilogue or
            # relu_bias_bwd of a layer with cout == stride; make it so otherwise instead of trusting the producer
            if cout != dy.shape[3]:
                dy[..., cout:].zero_()
        if train_b:
            self._pgrad(train_b, dbias)
        if a['residual']:
            if a['res_mode'] == 2:
                self._add_grad(a['residual'], ops.upsample2x_bwd(g, y.dt), lo)
            else:
                self._add_grad(a['residual'], g, lo)
        # frames of the input this gradient window reaches through the temporal taps
        pt = a['pads'][0]
        T = xin.T if xin.N == 1 else xin.t.shape[0]
        ilo, ihi = (max(0, lo - pt), min(T, lo + n + pt)) if xin.N == 1 else (0, xin.t.shape[0])
        x_win = xin.t[ilo:ihi]
        if (ilo, ihi) != (lo, lo + n):
            g_emb = torch.zeros((ihi - ilo,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            g_emb[lo - ilo:lo - ilo + n] = g
        else:
            g_emb = g
        Tw = (ihi - ilo) if xin.N == 1 else xin.T
        def build():
            w5 = self._master(a['w'])
            w5 = w5 if w5.dim() == 5 else w5.unsqueeze(2)
            scale = self._master(a['scale']) if a['scale'] else None
            return ops.ConvGrad(w5, scale, a['strides'], a['pads'], y.dt, xin.t.shape[3], g.shape[3])
        cg = self._conv_grad(i, build)
        if self._trainable(a['w']):
            gfr = (lo - ilo, n) if xin.N == 1 else None
            gt = self.gt_arena.get(a['w']) if (self.gt_arena is not None and a['w'] in self.arena) else None
            if a['w'] in self._sealed:
                self._pgrad(a['w'], None)       # raises: the bucket of this weight is already being exchanged
            job = None
            if gt is not None and cg.pointwise and self._pw_batch > 0:
                # pointwise layers: queued, and run as ONE grouped launch per gradient bucket / per `WGRAD_PW_BATCH` layers (a layer launched
                # alone is split over all CUs and adds 64 MB of partial tiles with float atomics; a shared grid a tenth of that)
                job = cg.weight_acc_job(x_win if x_win.is_contiguous() else x_win.contiguous(), g_emb, Tw, gt, g_frames=gfr)
            if job is not None:
                self._pw_pending.append((a['w'], job))
                self._deferred[a['w']] = cg.scale
                if len(self._pw_pending) >= self._pw_batch:
                    self._flush_pw()
            elif gt is not None and cg.weight_acc(x_win, g_emb, Tw, gt, g_frames=gfr):
                self._deferred[a['w']] = cg.scale
            else:
                dW, _ = cg.weight(x_win, g_emb, Tw, g_frames=gfr, out=self._pgrad_out(a['w']))
                self._pgrad(a['w'], dW)
        if op.inputs[0] not in self.no_grad:
            f, H, W, _ = xin.t.shape
            # a gradient of the same frame window that already waits for this input (the block's shortcut): the data-gradient conv
            # adds into it in its epilogue instead of producing a second tensor for an ATen add
            into = None
            pend = self.grads.get(op.inputs[0])
            if pend and len(pend) == 1:
                t0, l0, m0 = pend[0]
                assert not m0, 'a ReLU-masked gradient is the ONLY contribution of its blob (one reader): nothing may be added into it'
                if l0 == ilo and t0.shape[0] == ihi - ilo and t0.dtype == ops.tdtype(y.dt) and tuple(t0.shape[1:3]) == (H, W) and \
                        t0.is_contiguous() and not getattr(t0, '_roi_acc', False) and t0.shape[3] == ops.round_up(cg.cin, 64):
                    into = t0
            # ReLU backward of the input blob fused into this launch's epilogue: x = relu(...) is read by this conv only, its
            # gradient has this one contribution, and its producer is a conv this executor differentiates
            mask = None
            prod = self.net.producer(op.inputs[0])
            if (into is None and cfg.HIP.get('FUSE_RELU_BWD', True) and prod is not None and prod.type == 'Conv' and
                    isinstance(prod.args, dict) and prod.args.get('relu') and self._readers.get(op.inputs[0], 0) == 1 and
                    not pend and xin.keyframe is None and not xin.t2c and xin.t.shape[3] == ops.round_up(cg.cin, 64)):
                mask = x_win if x_win.is_contiguous() else None
            # a queued pointwise weight gradient may still have to READ the tensor the sum would be written into (the block output's gradient
            # is both branch2c's `g` and the shortcut contribution of the block input): then the sum goes to a new tensor
            # (byte-range overlap with BOTH operands of every queued job -- an offset view or the x operand would have slipped past
            #  an equality test of the g operand's start pointer)
            held = into is not None and any(_bytes_overlap(into, j[1]) or _bytes_overlap(into, j[2]) for _, j in self._pw_pending)
            # ... and when this conv is the LAST of the blob's readers to contribute (a residual block's output: read by the next block's
            # first conv and by its shortcut), the same epilogue applies the ReLU backward of the blob's producer to the finished sum
            # (dat_conv3d_fwd_sum_mask, round 6): that producer's elementwise mask pass (three passes over the block output) goes
            if (into is not None and cfg.HIP.get('FUSE_RELU_BWD', True) and cfg.HIP.get('FUSE_RELU_SUM_BWD', True) and _SUM_FUSE_ENV and prod is not None and
                    prod.type == 'Conv' and isinstance(prod.args, dict) and prod.args.get('relu') and
                    self._readers.get(op.inputs[0], 0) == 2 and self._ncontrib.get(op.inputs[0], 0) == 1 and
                    xin.keyframe is None and not xin.t2c and x_win.is_contiguous() and tuple(x_win.shape) == tuple(into.shape) and
                    x_win.dtype == into.dtype):
                mask = x_win
            dx = cg.data(g_emb, Tw, H, W, accumulate_into=into, g_frames=(lo - ilo, n) if xin.N == 1 else None, mask=mask, inplace=not held)
            if into is None:
                self._add_grad(op.inputs[0], dx, ilo, masked=mask is not None)
            else:
                self._ncontrib[op.inputs[0]] = self._ncontrib.get(op.inputs[0], 0) + 1
                if held or mask is not None:
                    self.grads[op.inputs[0]] = [(dx, ilo, mask is not None)]

    def _bwd_rpn_head(self, i):
        ws = self.ws
        lo, do, gi = self._fused[i]
        name = lo.outputs[0] + '+' + do.outputs[0]
        y = ws.blobs[name]
        dy, _lo = self._take_grad(name, y.dt)
        if dy is None:
            return
        xin = ws.blobs[lo.inputs[0]]
        A, D = lo.args['dim_out'], do.args['dim_out']
        dbias = torch.zeros(A + D, dtype=torch.float32, device=ws.device)
        g = ops.relu_bias_bwd(dy, y.t, y.dt, A + D, relu=False, dbias=dbias)
        w = None
        if xin.t2c or ('rpnhead', lo.args['w'], do.args['w'], int(xin.t.shape[3]), int(g.shape[3])) not in self._iter_cache:
            w = torch.cat([self._master(lo.args['w']).reshape(A, -1), self._master(do.args['w']).reshape(D, -1)], dim=0)
        if xin.t2c:
            # heads over time-moved-to-channels (FPN tube RPN, channel index t*C + c): one 1x1 weight slice per input frame;
            # the head has ONE output frame per clip, every input frame gets its own data gradient
            assert xin.N == 1, 'tube RPN training assumes one clip per forward (TRAIN.IMS_PER_BATCH 1)'
            T, C = xin.T, xin.C
            f, H, W, cs = xin.t.shape
            wT = w.view(A + D, T, C)
            dWs, dxs = [], []
            for t in range(T):
                cg = ops.ConvGrad(wT[:, t].reshape(A + D, C, 1, 1, 1).contiguous(), None, (1, 1), (0, 0, 0), y.dt, cs, g.shape[3])
                dW_t, _ = cg.weight(xin.t[t:t + 1], g, 1)
                dWs.append(dW_t.reshape(A + D, C))
                dxs.append(cg.data(g, 1, H, W))
            dW = torch.stack(dWs, dim=1).reshape(A + D, T * C)
            self._pgrad(lo.args['w'], dW[:A])
            self._pgrad(do.args['w'], dW[A:])
            self._pgrad(lo.args['b'], dbias[:A])
            self._pgrad(do.args['b'], dbias[A:])
            self._add_grad(lo.inputs[0], torch.cat(dxs, dim=0))
            return
        # (the levels above the first share the parameters: one ConvGrad -- one pack of the data-gradient weights -- per iteration)
        ck = ('rpnhead', lo.args['w'], do.args['w'], int(xin.t.shape[3]), int(g.shape[3]))
        cg = self._iter_cache.get(ck)
        if cg is None:
            cg = self._iter_cache[ck] = ops.ConvGrad(w.view(A + D, -1, 1, 1, 1), None, (1, 1), (0, 0, 0), y.dt, xin.t.shape[3], g.shape[3])
        dW, _ = cg.weight(xin.t, g, xin.T)
        self._pgrad(lo.args['w'], dW[:A])
        self._pgrad(do.args['w'], dW[A:])
        self._pgrad(lo.args['b'], dbias[:A])
        self._pgrad(do.args['b'], dbias[A:])
        f, H, W, _ = xin.t.shape
        self._add_grad(lo.inputs[0], cg.data(g, xin.T, H, W))

    def bwd_TimeToChannel(self, i, op):
        # a re-interpretation of the same [T, H, W, C] tensor: the gradient passes through unchanged
        x = self.ws.blobs[op.inputs[0]]
        dy, lo = self._take_grad(op.outputs[0], x.dt)
        if dy is not None:
            self._add_grad(op.inputs[0], dy, lo)

    def bwd_FC(self, i, op):
        ws, a = self.ws, op.args
        out = op.outputs[0]
        y = ws.blobs[out]
        dy, _lo = self._take_grad(out, y.dt)
        if dy is None:
            return
        x = ws.blobs[op.inputs[0]]
        # (the kernel ACCUMULATES the bias reduction: straight into the zeroed arena view when there is one -- no fill, no add)
        dbias = self.arena[a['b']] if (self.arena is not None and a['b'] in self.arena) else \
            torch.zeros(a['dim_out'], dtype=torch.float32, device=ws.device)
        g = ops.relu_bias_bwd(dy, y.t, y.dt, a['dim_out'], relu=a['relu'], dbias=dbias)
        self._pgrad(a['b'], dbias)
        w = self._master(a['w'])
        if x.kind == 'fmap':   # flattened RoI features: reference order (c, t, h, w), ours (t, h, w, c)
            f, p, p2, cs = x.t.shape
            xin = x.t.view(1, 1, f // x.T, x.T * p * p2 * cs)
            fwd = ws._layers.get((self.net.name, i))
            if isinstance(fwd, ops.ConvLayer) and fwd.w_src is not None and fwd.w_src.numel() == w.numel() and not fwd.is_dgrad:
                wk = fwd.w_src.view(w.shape[0], -1)     # the forward layer of this iteration holds the permuted copy already (51 MB for fc6)
            else:
                wk = w.view(w.shape[0], x.C, x.T, p, p2).permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1)
        else:
            xin, wk = x.t, w
        cin_real = wk.shape[1]
        if xin.shape[3] != cin_real:     # channel-padded rows: pad the weight columns
            wk = torch.nn.functional.pad(wk, (0, xin.shape[3] - cin_real))
        cg = ops.ConvGrad(wk.reshape(wk.shape[0], -1, 1, 1, 1).contiguous(), None, (1, 1), (0, 0, 0), y.dt, xin.shape[3],
                          g.shape[3])
        # a plain FC whose rows are not channel-padded: the kernel's finish writes the gradient where it belongs (first contribution of
        # the step); fc6 (permuted RoI features) and padded rows go through a temporary
        direct = self._pgrad_out(a['w']) if (x.kind != 'fmap' and xin.shape[3] == cin_real) else None
        dW, _ = cg.weight(xin, g, 1, out=direct.view(-1) if direct is not None else None)
        if direct is not None:
            self._pgrad(a['w'], direct)
        else:
            dW = dW.reshape(dW.shape[0], -1)[:, :cin_real]
            if x.kind == 'fmap':
                dW = dW.view(w.shape[0], x.T, p, p2, x.C).permute(0, 4, 1, 2, 3).reshape(w.shape)
            self._pgrad(a['w'], dW)
        if op.inputs[0] not in self.no_grad:
            dx = cg.data(g, 1, 1, xin.shape[2])
            if x.kind == 'fmap':
                dx = dx[..., :cin_real].reshape(x.t.shape)
            self._add_grad(op.inputs[0], dx)

    def bwd_ConvTranspose(self, i, op):
        ws, a = self.ws, op.args
        out = op.outputs[0]
        y = ws.blobs[out]
        dy, _lo = self._take_grad(out, y.dt)
        if dy is None:
            return
        x = ws.blobs[op.inputs[0]]
        if x.t2c:
            return self._bwd_deconv_over_time_channels(op, x, y, dy)
        K, Cin = a['dim_out'], a['dim_in']
        dbias4 = torch.zeros(4 * K, dtype=torch.float32, device=ws.device)
        g = ops.relu_bias_bwd(dy, y.t, y.dt, 4 * K, relu=False, dbias=dbias4)
        self._pgrad(a['b'], dbias4.view(4, K).sum(0))
        w = self._master(a['w'])                       # [Cin, K, 4, 4]
        w3 = ops.deconv_k4s2_as_conv3x3(w)             # [4K, Cin, 1, 3, 3]
        cg = ops.ConvGrad(w3, None, (1, 1), (0, 1, 1), y.dt, x.t.shape[3], g.shape[3])
        dW3, _ = cg.weight(x.t, g, 1)                  # [4K, Cin, 1, 3, 3]
        self._pgrad(a['w'], self._deconv_filter_grad(dW3, K, Cin))
        f, H, W, _ = x.t.shape
        self._add_grad(op.inputs[0], cg.data(g, 1, H, W))

    def _deconv_filter_grad(self, dW3, K, Cin):
        """Gradient of the sub-pixel conv weight [4K, Cin, 1, 3, 3] -> gradient of the ConvTranspose filter [Cin, K, 4, 4]: the transpose of
        the sub-pixel weight map (elementwise.hip deconv_k4s2_weights_kernel) -- every (ky, kx) of the 4x4 kernel appears exactly once,
        at sub-pixel (a, b) = ((ky+1)&1, (kx+1)&1), tap dy = (a + 1 - ky) / 2: one gather over index tables instead of sixteen strided
        copies into a zeroed tensor."""
        d4 = dW3.view(2, 2, K, Cin, 3, 3)
        idx = getattr(self, '_deconv_idx', None)
        if idx is None or idx[0].device != d4.device:
            ks = torch.arange(4)
            sub = (ks + 1) & 1                                              # sub-pixel a (rows) / b (columns) of kernel row / column k
            tap = torch.div(sub + 1 - ks, 2, rounding_mode='floor') + 1     # tap of the 3 x 3 conv that holds it
            AA, BB = sub.view(4, 1).expand(4, 4), sub.view(1, 4).expand(4, 4)
            TY, TX = tap.view(4, 1).expand(4, 4), tap.view(1, 4).expand(4, 4)
            idx = self._deconv_idx = tuple(t.contiguous().to(d4.device) for t in (AA, BB, TY, TX))
        AA, BB, TY, TX = idx
        dw = d4.permute(0, 1, 4, 5, 3, 2)[AA, BB, TY, TX]                  # [4, 4, Cin, K]
        return dw.permute(2, 3, 0, 1)

    def _bwd_deconv_over_time_channels(self, op, x, y, dy):
        """Backward of workspace.Executor._deconv_over_time_channels (the reference-default keypoint deconv over time-moved-to-channels,
        model_builder.py:765-767 / :848-856): the head has ONE output frame per roi and T input frames, so every input frame t gets its own
        data gradient through its own [4*T*K, C] filter slice -- T plain sub-pixel-conv gradients (the FPN tube RPN head does the same,
        _bwd_rpn_head).  With group = T only the diagonal [C, K] blocks are parameters: the gradient of block t is read off frame t's
        slice."""
        ws, a = self.ws, op.args
        T, C = x.T, x.C
        TK = a['dim_out']
        R = x.N
        dbias4 = torch.zeros(4 * TK, dtype=torch.float32, device=ws.device)
        g = ops.relu_bias_bwd(dy, y.t, y.dt, 4 * TK, relu=False, dbias=dbias4)
        self._pgrad(a['b'], dbias4.view(4, TK).sum(0))
        w = self._master(a['w'])
        group = self._deconv_group(a, w)
        dense = self.deconv_dense_filter(w, T, group)                   # [T*C, T*K, 4, 4]
        w3 = ops.deconv_k4s2_as_conv3x3(dense).view(4 * TK, T, C, 1, 3, 3)
        f, H, W, cs = x.t.shape
        xv = x.t.view(R, T, H, W, cs)
        dxs, dWs = [], []
        for t in range(T):
            cg = ops.ConvGrad(w3[:, t].contiguous(), None, (1, 1), (0, 1, 1), y.dt, cs, g.shape[3])
            dW_t, _ = cg.weight(xv[:, t].contiguous(), g, 1)            # [4TK, C, 1, 3, 3]
            dWs.append(self._deconv_filter_grad(dW_t, TK, C))           # [C, T*K, 4, 4]: rows t*C.. of the dense filter's gradient
            if op.inputs[0] not in self.no_grad:
                dxs.append(cg.data(g, 1, H, W))
        if group == 1:
            self._pgrad(a['w'], torch.cat(dWs, dim=0))
        else:
            k = TK // T
            self._pgrad(a['w'], torch.cat([dWs[t][:, t * k:(t + 1) * k] for t in range(T)], dim=0).contiguous())
        if dxs:
            self._add_grad(op.inputs[0], torch.stack(dxs, dim=1).reshape(f, H, W, dxs[0].shape[3]))

    def bwd_BilinearInterpolation(self, i, op):
        ws = self.ws
        d, _lo = self._take_grad(op.outputs[0], ops.F32)
        if d is None:
            return
        x = ws.blobs[op.inputs[0]]
        f, S, _, cs = x.t.shape
        K, up = op.args['dim'], op.args['up_scale']
        self._add_grad(op.inputs[0], ops.kps_finalize_bwd(d.contiguous(), x.dt, x.N, x.T, S, cs, K, up))

    def bwd_TubeDeltasToRows(self, i, op):
        """rows [R, K*T*4] -> head output [R*T, 1, 1, cs] with channels (k, xywh) per frame (model_builder.py:446-473)"""
        x = self.ws.blobs[op.inputs[0]]
        d, _lo = self._take_grad(op.outputs[0], ops.F32)
        if d is None:
            return
        R, T, cs = x.N, x.T, x.t.shape[3]
        K4 = x.C
        g = torch.zeros((R, T, cs), dtype=torch.float32, device=d.device)
        g[:, :, :K4] = d.view(R, K4 // 4, T, 4).permute(0, 2, 1, 3).reshape(R, T, K4)
        self._add_grad(op.inputs[0], g.view(R * T, 1, 1, cs))

    def bwd_TimeMean(self, i, op):
        """scores averaged over the T frames of each RoI: every frame receives d / T"""
        x = self.ws.blobs[op.inputs[0]]
        y = self.ws.blobs[op.outputs[0]]
        if y.kind != 'rows':
            return          # the RPN's TimeMean is folded into the proposal / loss kernels
        d, _lo = self._take_grad(op.outputs[0], ops.F32)
        if d is None:
            return
        R, T, cs = x.N, x.T, x.t.shape[3]
        g = torch.zeros((R, T, cs), dtype=torch.float32, device=d.device)
        g[:, :, :x.C] = (d.view(R, 1, -1)[:, :, :x.C] / float(T))
        self._add_grad(op.inputs[0], g.view(R * T, 1, 1, cs))

    def bwd_SpatialMean(self, i, op):
        x = self.ws.blobs[op.inputs[0]]
        d, _lo = self._take_grad(op.outputs[0], x.dt)
        if d is None:
            return
        f, h, w, cs = x.t.shape
        self._add_grad(op.inputs[0], (d.view(f, 1, 1, cs).float() / float(h * w)).to(d.dtype).expand(f, h, w, cs).contiguous())

    def bwd_TimePoolAvg(self, i, op):
        """BODY_HEAD_LINK 'avg' (detector.py:559-569): every frame receives d / T"""
        x = self.ws.blobs[op.inputs[0]]
        d, _lo = self._take_grad(op.outputs[0], x.dt)
        if d is None:
            return
        f, h, w, cs = x.t.shape
        g = (d.view(x.N, 1, h, w, cs).float() / float(x.T)).to(d.dtype).expand(x.N, x.T, h, w, cs).reshape(f, h, w, cs)
        self._add_grad(op.inputs[0], g.contiguous())

    def bwd_TimeToBatch(self, i, op):
        if op.outputs[0] in self.grads:
            self.grads.setdefault(op.inputs[0], []).extend(self.grads.pop(op.outputs[0]))

    bwd_Alias = bwd_TimeToBatch
    bwd_BatchToTime = bwd_TimeToBatch       # (views of the keypoint maps in the order they were written: the gradient passes through as it is)

    def bwd_RoIFeatureTransform(self, i, op):
        ws, a = self.ws, op.args
        y = ws.blobs[op.outputs[0]]
        dy, _lo = self._take_grad(op.outputs[0], y.dt)
        if dy is None:
            return
        names = op.inputs[:a['n_feat']]
        feats = [ws.blobs[n] for n in names]
        rois = ws.blobs[op.inputs[-1]]
        rt = rois.t.view(-1, rois.t.shape[-1]).float().contiguous()
        R = _count(rois) if rois.count is not None else rt.shape[0]
        Tr = (rt.shape[1] - 1) // 4
        accs = []
        for n, f in zip(names, feats):
            acc = None
            for t, _l, _m in self.grads.get(n, []):
                if t.dtype == torch.float32 and getattr(t, '_roi_acc', False):
                    acc = t
            if acc is None:
                acc = torch.zeros(f.t.shape, dtype=torch.float32, device=ws.device)
                acc._roi_acc = True
                self._add_grad(n, acc)
            accs.append(acc)
        ops.roi_align_bwd(accs, a['scales'], y.dt, rt[:R], dy, T=feats[0].T, Tr=Tr, t0=0, pooled=a['resolution'],
                          sampling=a['sampling_ratio'], k_min=cfg.FPN.ROI_MIN_LEVEL,
                          canon_scale=float(cfg.FPN.ROI_CANONICAL_SCALE), canon_level=cfg.FPN.ROI_CANONICAL_LEVEL)

    def bwd_SliceKeyFrame(self, i, op):
        x = self.ws.blobs[op.inputs[0]]
        dy, _lo = self._take_grad(op.outputs[0], x.dt)
        if dy is None:
            return
        k = op.args['keyframe']
        if x.N == 1:
            self._add_grad(op.inputs[0], dy, k)          # a one-frame window: no zero-filled T-frame tensor
            return
        g = torch.zeros_like(x.t)
        f, h, w, c = x.t.shape
        g.view(x.N, x.T, h, w, c)[:, k] = dy.view(x.N, h, w, c)
        self._add_grad(op.inputs[0], g)

    def bwd_MaxPool(self, i, op):
        """Only the P6 'sub-sampling' pool (kernel 1, stride 2; FPN3D.py:155-164) sits above the frozen trunk."""
        a = op.args
        x = self.ws.blobs[op.inputs[0]]
        dy, lo = self._take_grad(op.outputs[0], x.dt)
        if dy is None:
            return
        assert a['k'] == 1 and a['stride'] == 2 and a['pad'] == 0, 'max-pool backward above the frozen trunk: P6 only'
        g = torch.zeros((dy.shape[0],) + tuple(x.t.shape[1:]), dtype=x.t.dtype, device=x.t.device)
        g[:, ::2, ::2] = dy
        self._add_grad(op.inputs[0], g, lo)

    # ---- update ---------------------------------------------------------------------------------------------------------------
    def loss_values(self):
        return {k: float(v.item()) for k, v in self.losses.items()}


_SUM_FUSE_ENV = os.environ.get('DAT_FUSE_RELU_SUM_BWD', '1') != '0'       # experiment switch for same-box A/B runs (cfg.HIP.FUSE_RELU_SUM_BWD is the setting)


def _bytes_overlap(a, b):
    """Do the memory extents of two (dense) tensors intersect?"""
    a0, b0 = a.data_ptr(), b.data_ptr()
    return a0 < b0 + b.numel() * b.element_size() and b0 < a0 + a.numel() * a.element_size()


def no_grad_blobs(net):
    """Blobs no gradient is computed for: the network input and everything produced up to the last StopGradient marker (the frozen
    conv1 / res2 trunk, ResNet3D.py:273-274).  ONE definition for the executor's backward skip and the Trainer's completion order."""
    ng = {'data'}
    stop = [i for i, op in enumerate(net.ops) if op.type == 'StopGradient']
    if stop:
        for op in net.ops[:stop[-1] + 1]:
            ng.update(op.outputs)
    return ng


def param_ready_index(net, fused=None):
    """For every parameter blob a net's ops reference (`w`, `b`): the SMALLEST index of an op that uses it.  The backward pass runs
    the ops from the last to the first, so the parameter's gradient is final once the op of that index has been differentiated
    (shared weights -- the RPN conv of the five FPN levels -- complete with their first use).  fused: Executor._fused, {index the
    fused RPN head launch runs at: (logits op, deltas op, ...)} -- both heads' parameters complete at that index."""
    idx = {}
    pos = {id(op): i for i, op in enumerate(net.ops)}
    for i, op in enumerate(net.ops):
        a = op.args if isinstance(op.args, dict) else {}
        for key in ('w', 'b'):
            n = a.get(key)
            if isinstance(n, str) and n:
                idx[n] = min(idx.get(n, i), i)
    for first, grp in (fused or {}).items():
        for op in grp[:2]:
            for key in ('w', 'b'):
                n = op.args.get(key)
                if isinstance(n, str) and n:
                    idx[n] = min(idx.get(n, first), first, pos.get(id(op), first))
    # Parameters that never receive a gradient are final before the backward pass starts.  The rule is the EXECUTOR's own
    # (TrainExecutor.backward skips an op when all its outputs are in `no_grad_blobs`): a parameter is frozen when every op that uses
    # it is skipped -- not "last use before the marker", which would seal a parameter that a later op also uses.
    ng = no_grad_blobs(net)
    users = {}
    for i, op in enumerate(net.ops):
        a = op.args if isinstance(op.args, dict) else {}
        for key in ('w', 'b'):
            n = a.get(key)
            if isinstance(n, str) and n:
                users.setdefault(n, []).append(op)
    for n, us in users.items():
        if all(u.outputs and all(o in ng for o in u.outputs) for u in us):
            idx[n] = len(net.ops)
    return idx


class GradExchange(object):
    """The one exchange step of the path: sum all-reduce of the flat gradient buffer, bucket by bucket, each
    bucket started AS SOON AS its gradients are final while the backward pass of the earlier layers keeps the GPU busy -- what the
    reference gets from per-blob NCCLAllreduce ops scheduled by Caffe2's DAG executor (lib/modeling/model_builder.py:931-942).

    buckets: [(lo, hi)] element ranges of `flat` in the order the backward pass completes them.  `ready(k)` hands bucket k to the
    collective: on a CUDA buffer the reduce is enqueued on a COMMUNICATION stream behind an event of the compute stream (RCCL through
    torch.distributed, or the C ABI's dat_allreduce_bucket with cfg.HIP.RCCL_DIRECT); `finish()` starts what was not started, and
    makes the compute stream wait for the communication stream.  Backends: nccl (= RCCL over xGMI), gloo on a CPU buffer (tests), gloo
    on a CUDA buffer (two ranks sharing one GPU in the tests: staged through pinned host memory, synchronous).  overlap=False: nothing
    starts before finish() -- the serial exchange of rounds 1-3, kept as the A/B switch (cfg.HIP.OVERLAP_ALLREDUCE)."""

    def __init__(self, flat, buckets, dist, overlap=True, direct=None):
        self.flat, self.buckets, self.dist, self.overlap, self.direct = flat, list(buckets), dist, bool(overlap), direct
        self.cuda = bool(flat.is_cuda)
        self.backend = dist.get_backend() if dist is not None else None
        self.comm_stream = torch.cuda.Stream() if self.cuda else None
        self.stats = {}
        self._host = None
        self.begin()

    def begin(self):
        self.stats = {}             # (of THIS iteration: finish(timing=True) fills it; never an earlier iteration's numbers)
        self.started = [False] * len(self.buckets)
        self.order = []             # bucket indices in launch order (tests)
        self._works, self._events = [], []

    def ready(self, k):
        if self.overlap:
            self._start(k)

    def _start(self, k):
        if self.started[k]:
            return
        self.started[k] = True
        self.order.append(k)
        lo, hi = self.buckets[k]
        if hi <= lo:
            return
        t = self.flat[lo:hi]
        if not self.cuda:
            self._works.append(self.dist.all_reduce(t, async_op=True))
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        if self.backend == 'gloo':          # host-staged (tests: ranks that share one GPU cannot form an RCCL communicator)
            ev.synchronize()
            if self._host is None or self._host.numel() < t.numel():
                self._host = torch.empty(max(t.numel(), max(h - l for l, h in self.buckets)), dtype=t.dtype).pin_memory()
            h = self._host[:t.numel()]
            h.copy_(t)
            self.dist.all_reduce(h)
            t.copy_(h)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            e0.record(self.comm_stream)
            if self.direct is not None:
                self.direct.reduce_slice(self.flat, lo, hi - lo)
            else:
                w = self.dist.all_reduce(t, async_op=True)
                w.wait()                    # the COMMUNICATION stream waits for the collective (in order with the next bucket)
            e1.record(self.comm_stream)
        self._events.append((e0, e1))

    def finish(self, timing=False):
        """Start every bucket not started yet, then make the caller's stream wait for all of them.  timing=True (bench): synchronises
        and fills `stats` = {allreduce_ms: time the collectives ran, exposed_allreduce_ms: time the compute stream stood waiting}."""
        for k in range(len(self.buckets)):
            self._start(k)
        for w in self._works:
            w.wait()
        if self.cuda and self.backend != 'gloo':
            cur = torch.cuda.current_stream()
            a = b = None
            if timing:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(cur)
            done = torch.cuda.Event()
            done.record(self.comm_stream)
            cur.wait_event(done)
            if timing:
                b.record(cur)
                b.synchronize()
                self.stats = {'allreduce_ms': sum(e0.elapsed_time(e1) for e0, e1 in self._events),
                              'exposed_allreduce_ms': a.elapsed_time(b), 'buckets': len(self._events),
                              'bytes': 4 * sum(h - l for l, h in self.buckets)}


class Trainer(object):
    """One training process (one GPU): forward + backward + gradient all-reduce + momentum SGD on device fp32 masters
    (tools/train_net.py:120-170, model_builder.py:908-985).

    The trainable parameters, their momentum and their gradients each live in ONE flat fp32 buffer (weights first, then biases):
    the workspace's device masters are views into the first, the weight-gradient kernels write into views of the third, the
    all-reduce runs over bucket-sized slices of it (no flatten / unflatten copies) and the update is two `dat_sgd_momentum`
    launches (weights; biases with their 2x learning rate and no decay).  Packed layer weights are refreshed in place after the
    update; layers of frozen parameters are left alone."""

    BUCKET_BYTES = 64 << 20      # xGMI rings are per-link bound: few, large buckets
    TAIL_BYTES = 8 << 20         # ... except the one that cannot overlap anything (see __init__)

    def __init__(self, model, ws, dist=None):
        self.model, self.ws, self.dist = model, ws, dist
        from detectandtrack_amd.workspace import _x3
        assert not _x3(ws), "cfg.HIP.DTYPE 'bf16x3' is an inference mode (train in 'bf16' or 'fp32')"
        self.trainable = list(model.TrainableParams())
        self.biases = set(model.biases)
        self.iter = 0
        # Flat order = the order in which the backward pass COMPLETES the gradients (heads first, res3 last; weights, then biases):
        # a gradient bucket is then one contiguous slice that becomes final as a whole, early, and can be exchanged while the
        # backward of the earlier layers still runs (GradExchange).
        planner = TrainExecutor(ws, model.net)
        planner._plan_rpn_siblings()
        self.ready_index = param_ready_index(model.net, planner._fused)
        n_ops = len(model.net.ops)
        rank_of = {n: i for i, n in enumerate(self.trainable)}
        by_done = sorted(self.trainable, key=lambda n: (-self.ready_index.get(n, -1), rank_of[n]))   # unreferenced parameters last
        order = [n for n in by_done if n not in self.biases] + [n for n in by_done if n in self.biases]
        sizes = [int(np.prod(ws.params[n].shape)) for n in order]
        self.n_weights = sum(sz for n, sz in zip(order, sizes) if n not in self.biases)
        total = sum(sizes)
        dev = ws.device
        self.flat_w = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.momentum, self.arena = {}, {}
        self.flat_gt = torch.zeros(self.n_weights, dtype=torch.float32, device=dev) if cfg.HIP.get('DEFER_WGRAD_FINISH', True) else None
        self.gt_arena = {} if self.flat_gt is not None else None
        off = 0
        for n, sz in zip(order, sizes):
            shape = tuple(ws.params[n].shape)
            view = self.flat_w[off:off + sz].view(shape)
            view.copy_(ws.dev_param(n))
            ws._dev_params[n] = view                      # the workspace's device master IS the slice of the flat buffer
            self.momentum[n] = self.flat_v[off:off + sz].view(shape)
            self.arena[n] = self.flat_g[off:off + sz].view(shape)
            if self.gt_arena is not None and n not in self.biases and len(shape) >= 4:
                self.gt_arena[n] = self.flat_gt[off:off + sz]      # (weights come first in the flat order: off < n_weights)
            off += sz
        ws._layers.clear()                                # layers built before held the old master tensors
        self._train_ptrs = {ws._dev_params[n].data_ptr() for n in order}
        # gradient buckets: consecutive parameters of the flat order up to BUCKET_BYTES; all biases (a few KB) are the last bucket.
        # ready_at = the op index after whose backward every gradient of the bucket is final
        self.buckets, cur, cur_lo, off = [], [], 0, 0
        per = max(1, self.BUCKET_BYTES // 4)
        for n, sz in zip(order, sizes):
            if cur and (n in self.biases) != (cur[0] in self.biases or False):
                self.buckets.append((cur_lo, off, cur))
                cur, cur_lo = [], off
            cur.append(n)
            off += sz
            if n not in self.biases and off - cur_lo >= per:
                self.buckets.append((cur_lo, off, cur))
                cur, cur_lo = [], off
        if cur:
            self.buckets.append((cur_lo, off, cur))
        # the LAST weight bucket completes with the backward pass itself, so its exchange is the exposed one: keep it small by cutting
        # it where the trailing part (the parameters that complete last: res3) drops under TAIL_BYTES
        wb = [k for k, (_, _, names) in enumerate(self.buckets) if names[0] not in self.biases]
        if wb:
            lo, hi, names = self.buckets[wb[-1]]
            szs = [int(np.prod(ws.params[n].shape)) for n in names]
            tail, cut = 0, len(names)
            while cut > 1 and (tail + szs[cut - 1]) * 4 <= self.TAIL_BYTES:
                cut -= 1
                tail += szs[cut]
            if (hi - lo) * 4 > self.TAIL_BYTES and 0 < tail < hi - lo and cut < len(names):
                self.buckets[wb[-1]:wb[-1] + 1] = [(lo, hi - tail, names[:cut]), (hi - tail, hi, names[cut:])]
        self.bucket_ready_at = [min(self.ready_index.get(n, -1) for n in names) for _, _, names in self.buckets]
        self.exchange = None
        self.last_exchange_stats = {}

    def _check_aliases(self):
        """The workspace's device masters of the trainable parameters must still BE the slices of `flat_w` (Workspace.set_param copies
        in place; anything that replaced a master would make the update silently train a buffer nobody reads)."""
        lo, hi = self.flat_w.data_ptr(), self.flat_w.data_ptr() + 4 * self.flat_w.numel()
        for n in self.trainable:
            t = self.ws._dev_params.get(n)
            if t is None or not (lo <= t.data_ptr() < hi):
                raise RuntimeError('device master of %r no longer aliases the Trainer\'s flat weight buffer (parameter replaced after '
                                   'the Trainer was built): rebuild the Trainer' % n)

    def step(self, lr, zero_grad=True, update=True, timing=False):
        """One training iteration: forward + losses + backward (+ gradient exchange over the ranks) + momentum SGD.
        zero_grad=False adds this clip's gradients to the ones already in the buffer, update=False stops before the exchange and
        the update: `step(lr, update=False); step(lr, zero_grad=False)` is one iteration over two clips on one rank -- by linearity
        what two ranks with one clip each compute (losses carry the reference's 1 / NUM_GPUS, model_builder.py:484,625,884)."""
        ws = self.ws
        self._check_aliases()
        if getattr(ws, 'param_epoch', 0) != getattr(self, '_param_epoch', None):
            self._pack_sig = None        # layers were rebuilt since the last step: the cached pack table holds stale pointers
            self._param_epoch = getattr(ws, 'param_epoch', 0)
        if zero_grad:
            self.flat_g.zero_()
        if self.flat_gt is not None:
            self.flat_gt.zero_()
        ex = TrainExecutor(ws, self.model.net, arena=self.arena, gt_arena=self.gt_arena)
        ex.accumulate_all = not zero_grad
        ex.run()
        # (DAT_FORCE_EXCHANGE=1: run the exchange machinery with ONE rank too -- a sum over one rank is the identity -- so that the RCCL path,
        #  its communication stream and events can be exercised on a one-GPU box: tests/test_gpu_train.py, bench.py --mode train)
        multi = self.dist is not None and update and (self.dist.get_world_size() > 1 or os.environ.get('DAT_FORCE_EXCHANGE', '0') == '1')
        if multi:
            # every rank reduces the same flat buffer: a parameter without a gradient on this rank contributes its zeros
            xch = self._exchange()
            xch.begin()
            nxt = [0]

            def on_op_done(i):
                while nxt[0] < len(self.buckets) and i <= self.bucket_ready_at[nxt[0]]:
                    k = nxt[0]
                    if xch.overlap:
                        names = self.buckets[k][2]
                        ex._finish_deferred(only=set(names))       # this bucket's deferred weight gradients -> the flat buffer
                        ex.seal(names)
                        xch.ready(k)
                    nxt[0] += 1
            ex.backward(on_op_done)
            xch.finish(timing=timing)
            self.last_exchange_stats = dict(xch.stats)
        else:
            ex.backward()
        if not update:
            return ex
        nw, n = self.n_weights, self.flat_w.numel()
        # a learning-rate step rescales the update history (detector.py:606-640; a few times per schedule: one elementwise pass)
        corr = lr_policy.momentum_correction(getattr(self, '_fed_lr', None), float(np.float32(lr)))
        if corr is not None:
            self.flat_v.mul_(float(corr))
        self._fed_lr = float(np.float32(lr))            # (the reference's `lr` blob is float32)
        if nw > 0:
            ops.sgd_momentum(self.flat_w[:nw], self.flat_v[:nw], self.flat_g[:nw], lr, cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY, False)
        if n > nw:
            ops.sgd_momentum(self.flat_w[nw:], self.flat_v[nw:], self.flat_g[nw:], lr, cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY, True)
        self._refresh_layers()
        self.iter += 1
        return ex

    def _exchange(self):
        if self.exchange is None:
            direct = None
            if cfg.HIP.get('RCCL_DIRECT', False) and self.flat_g.is_cuda:
                # the same exchange through the C ABI (dat_allreduce_bucket); torch.distributed only carries the 128-byte communicator id
                direct = ops.BucketAllReduce(self.dist.get_rank(), self.dist.get_world_size())
            self.exchange = GradExchange(self.flat_g, [(lo, hi) for lo, hi, _ in self.buckets], self.dist,
                                         overlap=bool(cfg.HIP.get('OVERLAP_ALLREDUCE', True)), direct=direct)
        return self.exchange

    def _refresh_layers(self):
        """Packed weights follow the updated masters: layers (and cached gradient layers) packed straight from a trainable master are
        re-packed in place, layers of frozen parameters stay, layers built from derived tensors (concatenated RPN heads, permuted FC
        weights, the sub-pixel deconv) are dropped and rebuilt on the next forward."""
        ws = self.ws
        all_ptrs = {t.data_ptr() for t in ws._dev_params.values()}
        todo = []               # ConvLayers to re-pack: ONE batched launch (they were ~100 launches, 1.75 ms of a 23 ms iteration)
        for key in list(ws._layers.keys()):
            layer = ws._layers[key]
            src = getattr(layer, 'w_src', None) if isinstance(layer, ops.ConvLayer) else \
                (layer.w if isinstance(layer, ops.ConvGrad) else None)
            if src is None:     # anchors (tensors) stay; the fused stem holds conv1, frozen below the StopGradient marker in every config
                if isinstance(layer, ops.StemConv) and 'conv1_w' in self.trainable:
                    del ws._layers[key]
                continue
            ptr = src.data_ptr()
            if ptr in self._train_ptrs:
                if isinstance(layer, ops.ConvLayer):
                    todo.append(layer)
                elif layer._data_layer is not None:
                    todo.append(layer._data_layer)
            elif ptr not in all_ptrs:
                del ws._layers[key]
        if todo:
            sig = tuple(id(l) for l in todo)
            if getattr(self, '_pack_sig', None) != sig:
                self._pack_sig, self._pack_batch = sig, ops.PackBatch(todo)
            self._pack_batch.run()

    def momentum_blobs(self):
        """name -> host array of the momentum buffers (saved as `<param>_momentum`, reference utils/net.py:268-275)."""
        return {n: m.detach().cpu().numpy() for n, m in self.momentum.items()}

    def load_momentum(self, blobs):
        """Restore momentum buffers read from a checkpoint (utils.net.initialize_from_weights_file(momentum=...))."""
        for n, m in (blobs or {}).items():
            if n in self.momentum:
                self.momentum[n].copy_(torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)).to(self.flat_v.device).view_as(self.momentum[n]))

    def _all_reduce(self):
        """Serial bucketed sum all-reduce over slices of the flat gradient buffer (losses are already divided by NUM_GPUS,
        model_builder.py:932-942): the exchange without overlap, BUCKET_BYTES per collective."""
        per = max(1, self.BUCKET_BYTES // 4)
        n = self.flat_g.numel()
        x = GradExchange(self.flat_g, [(off, min(off + per, n)) for off in range(0, n, per)], self.dist, overlap=False,
                         direct=(ops.BucketAllReduce(self.dist.get_rank(), self.dist.get_world_size())
                                 if cfg.HIP.get('RCCL_DIRECT', False) and self.flat_g.is_cuda else None))
        x.finish()
