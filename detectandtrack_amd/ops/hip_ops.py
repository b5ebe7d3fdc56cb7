"""Thin Python wrappers over the C ABI (include/dat_hip.h) working on torch CUDA tensors.

torch is plumbing only: it owns HBM allocations and the HIP stream; every computation
below is a libdat_hip kernel.  No CPU fallbacks exist — a missing library or a missing
GPU raises.
"""
import ctypes as C

import numpy as np
import torch

from .. import libdat as L

F32, BF16 = L.DAT_F32, L.DAT_BF16


# the 16-bit element format of the LOADED library build (libdat.H16: 'bf16' = libdat_hip.so, 'fp16' = libdat_hip_f16.so, the same sources
# compiled with -DDAT_H16_IS_FP16): the C ABI's DAT_BF16 tag then means IEEE half, and the torch tensors that carry it are float16
H16_DTYPE = torch.float16 if L.H16 == 'fp16' else torch.bfloat16


def tdtype(dt):
    return H16_DTYPE if dt == BF16 else torch.float32


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream_handle(device=None):
    """Raw hipStream_t of torch's current stream on `device`.  `torch.cuda.current_stream()` builds a Stream object through four
    Python layers (8 us; two calls per launch were 3.6 ms of host time per training iteration of ~220 launches): the raw query
    is one C call."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device() if device is None else device)
    return torch.cuda.current_stream().cuda_stream


def _stream():
    return C.c_void_p(_stream_handle())


def round_up(v, m):
    return (v + m - 1) // m * m


_CTX = {}


def ctx(device=None):
    """Context of the current (device, HIP stream) pair, created lazily.  The C-ABI context owns scratch memory that
    kernels of ONE stream may use at a time, so concurrent streams (clip pipelining) get their own context."""
    if device is None:
        device = torch.cuda.current_device()
    key = (device, _stream_handle(device))
    if key not in _CTX:
        _CTX[key] = L.Ctx(device)
    return _CTX[key]


def drop_ctx(device=None):
    """Forgets (and thereby destroys) the context of the current (device, stream) pair; the next launch on it creates a new one, which
    reads the library's environment switches again.  The stream must be idle.  (torch hands out pooled stream handles: a test that
    wants a context made under a particular environment must not inherit one cached for an earlier stream with the same handle.)"""
    if device is None:
        device = torch.cuda.current_device()
    _CTX.pop((device, _stream_handle(device)), None)


def ws_info():
    """(device pointer, bytes, growth count) of the scratch buffer owned by the context of the current (device, stream)
    (dat_ws_info).  Growth retires the outgrown buffer instead of freeing it, so captured hipGraphs stay replayable."""
    p, n, g = C.c_void_p(), C.c_size_t(), C.c_int()
    ctx().check(L._lib.dat_ws_info(ctx().h, C.byref(p), C.byref(n), C.byref(g)))
    return (p.value or 0), int(n.value), int(g.value)


def ws_reserve(nbytes):
    """Grow the current context's scratch to at least `nbytes` now (dat_ws_reserve)."""
    ctx().check(L._lib.dat_ws_reserve(ctx().h, C.c_size_t(int(nbytes))))


def tune_plan(positions_per_block, ksplit):
    """dat_conv3d_tune_plan on the context of the current (device, stream): forces the launch plan of the conv launches that
    follow on THAT context (0, 0 = back to the makespan model).  Returns the C-ABI return code."""
    return L._lib.dat_conv3d_tune_plan(ctx().h, int(positions_per_block), int(ksplit))


def persistent_share(percent):
    """dat_conv3d_persistent_share on the context of the current (device, stream): the share of the CUs the persistent HBM-bound conv
    kernels launched on THAT context take (core/pipeline.py lowers it while several forwards are in flight)."""
    ctx().call('dat_conv3d_persistent_share', int(percent))


# ---- toy + AffineChannelNd (the reference's own native ops) ---------------------------------------
def zero_even(x):
    """In-place ZeroEven on a 1-D fp32 tensor (lib/ops/zero_even_op.cc:17-30 semantics)."""
    if x.dim() != 1:
        raise L.DatError('ZeroEven: X.ndim() == 1 required, got %d dims' % x.dim())
    ctx().call('dat_zero_even_fwd', _stream(), _ptr(x), C.c_longlong(x.numel()))
    return x


def affine_channel_nd(x, scale, bias, out=None):
    """y = x*scale[c] + bias[c] on NC(...) fp32; out may alias x (in-place)."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = x.new_empty(x.shape) if out is None else out
    n, c = x.shape[0], x.shape[1]
    inner = int(np.prod(x.shape[2:])) if x.dim() > 2 else 1
    ctx().call('dat_affine_channel_nd_fwd', _stream(), _ptr(x), _ptr(scale), _ptr(bias), _ptr(out), n, c,
               C.c_longlong(inner))
    return out


def affine_channel_nd_grad(dy, scale):
    dx = dy.new_empty(dy.shape)
    n, c = dy.shape[0], dy.shape[1]
    inner = int(np.prod(dy.shape[2:])) if dy.dim() > 2 else 1
    ctx().call('dat_affine_channel_nd_bwd', _stream(), _ptr(dy), _ptr(scale), _ptr(dx), n, c, C.c_longlong(inner))
    return dx


# ---- boundary layout moves ---------------------------------------------------------------------------
def to_ndhwc(x, dtype, cs=None):
    """x fp32 NC(T)HW (4-D or 5-D) -> [N*T, H, W, cs] in dtype."""
    if x.dim() == 4:
        x = x[:, :, None]
    x = x.contiguous()
    n, c, t, h, w = x.shape
    cs = cs or round_up(c, 64)
    out = torch.empty((n * t, h, w, cs), dtype=tdtype(dtype), device=x.device)
    ctx().call('dat_ncdhw_to_ndhwc', _stream(), _ptr(x), _ptr(out), dtype, n, c, t, h, w, cs)
    return out


def to_ncdhw(x, dtype, n, c, t):
    """x [N*T, H, W, cs] -> fp32 [N, C, T, H, W]."""
    f, h, w, cs = x.shape
    assert f == n * t
    out = torch.empty((n, c, t, h, w), dtype=torch.float32, device=x.device)
    ctx().call('dat_ndhwc_to_ncdhw', _stream(), _ptr(x), dtype, _ptr(out), n, c, t, h, w, cs)
    return out


def copy_frames(src, src_idx, dst, dst_idx):
    """dst[dst_idx[i]] = src[src_idx[i]] over whole frames (leading axis) of two contiguous tensors with equal frame size (dat_copy_frames)."""
    n = len(src_idx)
    assert n == len(dst_idx) and src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype
    fb = src[0].numel() * src.element_size()
    assert fb == dst[0].numel() * dst.element_size() and max(src_idx) < src.shape[0] and max(dst_idx) < dst.shape[0]
    ctx().call('dat_copy_frames', _stream(), _ptr(src), (C.c_int * n)(*[int(v) for v in src_idx]), _ptr(dst),
               (C.c_int * n)(*[int(v) for v in dst_idx]), n, C.c_longlong(fb))
    return dst


# ---- fused conv ------------------------------------------------------------------------------------------
class ConvLayer(object):
    """A conv with packed weights + fused epilogue parameters, ready to launch.

    w: fp32 [Cout, Cin, KT, KH, KW] (reference blob layout, CUDA tensor)
    scale/bias: fp32 [Cout] or None (AffineChannelNd / conv bias); padded to the stored Cout.
    """

    def __init__(self, w, scale=None, bias=None, stride=(1, 1), pads=(0, 0, 0), relu=False, dtype=BF16,
                 cin_stride=None, dgrad_of=None, x3=False):
        """dgrad_of = (w_fwd, scale_fwd): this layer is the DATA-GRADIENT conv of a forward conv with master weights w_fwd
        [CoutF, CinF, KT, KH, KW]; `w` is then ignored and the packed weights come straight from w_fwd (channels swapped, kernel
        flipped, AffineChannelNd scale folded in: dat_conv3d_pack_weights_dgrad)."""
        if dgrad_of is not None:
            w_fwd = dgrad_of[0].contiguous().float()
            self.w_src, self.dgrad_scale = w_fwd, (None if dgrad_of[1] is None else dgrad_of[1].contiguous().float())
            coutf, cinf = int(w_fwd.shape[0]), int(w_fwd.shape[1])
            self.cout_real, self.cin_real = cinf, coutf
            self.kt, self.kh, self.kw = [int(v) for v in w_fwd.shape[2:]]
            w = w_fwd
        else:
            w = w.contiguous().float()
            self.w_src, self.dgrad_scale = w, None          # kept (no copy when `w` already was a contiguous fp32 master): repack()
            self.cout_real, self.cin_real, self.kt, self.kh, self.kw = [int(v) for v in w.shape]
        self.is_dgrad = dgrad_of is not None
        self.dtype = dtype
        # x3 (cfg.HIP.DTYPE 'bf16x3'): fp32 activations, the conv itself on hi / lo bf16 splits of both operands -- three bf16 MFMAs per
        # k-slice, fp32 accumulate, ~2^-16 relative: the parity bar (kps_score < 1e-3) at several times the fp32-MFMA rate
        self.x3 = bool(x3)
        assert not self.x3 or (dtype == F32 and dgrad_of is None), 'bf16x3 is an inference mode over fp32 activations'
        self.cin = cin_stride or round_up(self.cin_real, 64)
        self.cout = round_up(self.cout_real, 4)
        self.cstride = round_up(self.cout_real, 64)  # outputs feed later convs: keep C % 64 == 0
        self.stride = tuple(stride)
        self.pads = tuple(pads)
        self.relu = bool(relu)
        dev = w.device
        self.scale = None
        self.bias = None
        if scale is not None:
            self.scale = torch.ones(self.cout, dtype=torch.float32, device=dev)
            self.scale[:self.cout_real] = scale.float()
        self.bias_src = None
        if bias is not None:
            if bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == self.cout:
                self.bias = bias            # the master itself: always current, nothing to refresh after an update
            else:
                self.bias = torch.zeros(self.cout, dtype=torch.float32, device=dev)
                self.bias[:self.cout_real] = bias.float()
                self.bias_src = bias
        d = self.desc(1, 1, 8, 8)
        nbytes = L.lib().dat_conv3d_packed_weight_bytes(C.byref(d))
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.repack(weights_only=True)

    def _x3_master(self):
        """[Cout, 3 * cin, KT, KH, KW] fp32: per 64-channel chunk of the (padded) input channels the block [W_hi | W_lo | W_hi],
        W_hi = bf16(W), W_lo = bf16(W - W_hi) -- met in the kernel by the input chunks [x_hi | x_hi | x_lo]."""
        w = self.w_src
        co, ci = int(w.shape[0]), int(w.shape[1])
        wp = torch.zeros((co, self.cin) + tuple(w.shape[2:]), dtype=torch.float32, device=w.device)
        wp[:, :ci] = w
        hi = wp.to(torch.bfloat16).float()
        lo = (wp - hi).to(torch.bfloat16).float()
        q = self.cin // 64
        shp = (co, q, 64) + tuple(w.shape[2:])
        m = torch.stack([hi.view(shp), lo.view(shp), hi.view(shp)], dim=2)        # [co, q, 3, 64, ...]
        return m.reshape((co, 3 * self.cin) + tuple(w.shape[2:])).contiguous()

    def repack(self, weights_only=False):
        """(Re-)derive the packed weights (and a padded bias copy) from the fp32 masters -- after an SGD update in place."""
        d = self.desc(1, 1, 8, 8)
        if self.x3:
            d.dtype, d.Cin = BF16, 3 * self.cin
            ctx().call('dat_conv3d_pack_weights', _stream(), C.byref(d), _ptr(self._x3_master()), self.cout_real, 3 * self.cin,
                       _ptr(self.packed))
        elif self.is_dgrad:
            ctx().call('dat_conv3d_pack_weights_dgrad', _stream(), C.byref(d), _ptr(self.w_src), self.cin_real, self.cout_real,
                       _ptr(self.dgrad_scale), _ptr(self.packed))
        else:
            ctx().call('dat_conv3d_pack_weights', _stream(), C.byref(d), _ptr(self.w_src), self.cout_real, self.cin_real,
                       _ptr(self.packed))
        if not weights_only and self.bias_src is not None:
            self.bias[:self.cout_real] = self.bias_src.float()

    def desc(self, frames, T, H, W, res_mode=0, relu=None, cstride=None, out_t=None, in_t=None):
        d = L.ConvDesc()
        d.dtype = L.DAT_BF16X3 if self.x3 else self.dtype
        d.frames, d.T, d.H, d.W, d.Cin = frames, T, H, W, self.cin
        d.Cout = self.cout
        d.out_cstride = cstride or self.cstride
        d.KT, d.KH, d.KW = self.kt, self.kh, self.kw
        d.stride_h, d.stride_w = self.stride
        d.pad_t, d.pad_h, d.pad_w = self.pads
        d.relu = int(self.relu if relu is None else relu)
        d.res_mode = res_mode
        d.out_t0, d.out_tn = out_t if out_t is not None else (0, 0)
        d.in_t0, d.in_tn = in_t if in_t is not None else (0, 0)
        return d

    def out_hw(self, H, W):
        return ((H + 2 * self.pads[1] - self.kh) // self.stride[0] + 1,
                (W + 2 * self.pads[2] - self.kw) // self.stride[1] + 1)

    def flops(self, frames, H, W):
        ho, wo = self.out_hw(H, W)
        return 2.0 * self.cout_real * self.cin_real * self.kt * self.kh * self.kw * frames * ho * wo

    def hbm_bytes(self, frames, H, W, oframes=None, res_mode=0):
        """Algorithmic HBM bytes of one launch: input + packed weights + output (+ residual), each touched once."""
        es = 2 if self.dtype == BF16 else 4
        ho, wo = self.out_hw(H, W)
        in_frames = frames if oframes is None else min(frames, oframes * self.kt)   # key-frame outputs read kt frames
        oframes = frames if oframes is None else oframes
        in_pos = H * W
        if self.kh == 1 and self.kw == 1:        # a strided 1x1 conv touches only the sampled positions of its input
            in_pos = ho * wo
        b = in_frames * in_pos * self.cin * es + self.kt * self.kh * self.kw * self.cout * self.cin * es
        b += oframes * ho * wo * self.cstride * es
        if res_mode == 1:
            b += oframes * ho * wo * self.cstride * es
        elif res_mode == 2:
            b += oframes * (ho // 2) * (wo // 2) * self.cstride * es
        return float(b)

    def __call__(self, x, T=1, residual=None, res_mode=None, out=None, out_t=None, in_t=None, x_split=None, zero_pad=True,
                 want_split=False, addend=None):
        """out_t = (t0, n): only output frames t0..t0+n-1 of every clip are computed and stored.
        x_split (bf16x3 layers): the hi / lo split of `x` when the caller already has it (a blob read by several convs is split once)."""
        frames, H, W, cin = x.shape
        assert cin == self.cin, 'channel stride %d != layer Cin %d' % (cin, self.cin)
        assert x.dtype == tdtype(self.dtype) and x.is_contiguous()
        if res_mode is None:
            res_mode = 1 if residual is not None else 0
        d = self.desc(frames, T, H, W, res_mode, out_t=out_t, in_t=in_t)
        ho, wo = self.out_hw(H, W)
        oframes = frames if out_t is None else frames // T * out_t[1]
        if out is None:
            # channels [cout, cstride) of the output are never written by the kernel: zero them when a later conv / FC reads the whole
            # channel stride against zero-padded weights (0 x garbage may be NaN); zero_pad=False: every reader stops at the real channels
            # (RPN head -> proposal kernels, cls_score / bbox_pred -> softmax / box decode, the deconv -> kps_finalize) -- no fill launch
            alloc = torch.zeros if (self.cstride != self.cout and zero_pad) else torch.empty
            out = alloc((oframes, ho, wo, self.cstride), dtype=x.dtype, device=x.device)
        if res_mode == 4:       # out = residual > 0 ? conv + addend : 0 (training: dat_conv3d_fwd_sum_mask; `addend` may be `out`)
            assert not self.x3 and addend is not None and residual is not None
            assert addend.shape == out.shape == residual.shape and addend.dtype == out.dtype == residual.dtype
            assert addend.is_contiguous() and residual.is_contiguous() and out.is_contiguous()
            ctx().call('dat_conv3d_fwd_sum_mask', _stream(), C.byref(d), _ptr(x), _ptr(self.packed), _ptr(self.scale),
                       _ptr(self.bias), _ptr(addend), _ptr(residual), _ptr(out))
            return out
        if not self.x3:
            ctx().call('dat_conv3d_fwd', _stream(), C.byref(d), _ptr(x), _ptr(self.packed), _ptr(self.scale),
                       _ptr(self.bias), _ptr(residual), _ptr(out))
            return out
        # bf16x3: the conv reads the hi / lo split of x; want_split: its epilogue also writes the split of ITS output (out._split) for the
        # convs that read it -- no pre-pass for them (dat_conv3d_fwd_x3; needs an unpadded channel stride)
        xin = x_split if x_split is not None else split_bf16x2(x)
        ysplit = None
        if want_split and self.cstride == self.cout and self.cstride % 64 == 0:
            ysplit = torch.empty(tuple(out.shape[:-1]) + (2 * self.cstride,), dtype=torch.bfloat16, device=out.device)
        ctx().call('dat_conv3d_fwd_x3', _stream(), C.byref(d), _ptr(xin), _ptr(self.packed), _ptr(self.scale),
                   _ptr(self.bias), _ptr(residual), _ptr(out), _ptr(ysplit))
        out._split = ysplit
        return out


def split_bf16x2(x):
    """fp32 [..., C] (C % 64 == 0, contiguous) -> the hi / lo bf16 split [..., 2C] the bf16x3 conv mode reads (dat_split_bf16x2)."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 64 == 0
    cs = int(x.shape[-1])
    out = torch.empty(tuple(x.shape[:-1]) + (2 * cs,), dtype=torch.bfloat16, device=x.device)
    ctx().call('dat_split_bf16x2', _stream(), _ptr(x), _ptr(out), C.c_longlong(x.numel() // cs), cs)
    return out


class ConvGrad(object):
    """Backward of a ConvLayer-shaped conv (Caffe2 ConvGradient via AddGradientOperators,
    model_builder.py:908-952).  `w`: the forward weight fp32 [Cout, Cin, KT, KH, KW]; `scale`: the fused
    AffineChannelNd scale (or None).  With z = conv(x, w), y = act(z*scale + bias + res) and g = dL/dy masked by the ReLU:
      data(g)   -> dL/dx  = conv_stride1(zero_insert(g), flip(w)^T * scale)      (the forward MFMA kernel)
      weight(x, g) -> G = dL/d(z) / scale-free weight gradient;  dL/dw = scale*G,  dL/dscale[c] = <w[c], G[c]>.
    """

    def __init__(self, w, scale, stride, pads, dtype, x_cstride, g_cstride):
        self.w = w.contiguous().float()
        self.cout, self.cin, self.kt, self.kh, self.kw = [int(v) for v in self.w.shape]
        self.scale = None if scale is None else scale.float()
        self.stride, self.pads, self.dtype = tuple(stride), tuple(pads), dtype
        self.x_cstride, self.g_cstride = x_cstride, g_cstride
        assert self.stride[0] == self.stride[1] and self.stride[0] in (1, 2)
        self._data_layer = None

    def _fwd_desc(self, frames, T, H, W):
        d = L.ConvDesc()
        d.dtype = self.dtype
        d.frames, d.T, d.H, d.W, d.Cin = frames, T, H, W, self.x_cstride
        d.Cout = round_up(self.cout, 4)
        d.out_cstride = self.g_cstride
        d.KT, d.KH, d.KW = self.kt, self.kh, self.kw
        d.stride_h, d.stride_w = self.stride
        d.pad_t, d.pad_h, d.pad_w = self.pads
        d.relu, d.res_mode, d.out_t0, d.out_tn, d.in_t0, d.in_tn = 0, 0, 0, 0, 0, 0
        return d

    def weight(self, x, g, T, want_dscale=False, g_frames=None, out=None):
        """x [frames,H,W,x_cstride], g [frames,Ho,Wo,g_cstride] -> (dW fp32 [Cout,Cin,KT,KH,KW] (already x scale),
        dscale fp32 [Cout] | None); out: a contiguous fp32 tensor of the weight's shape to write dW into (gradient arena)"""
        frames, H, W, _ = x.shape
        d = self._fwd_desc(frames, T, H, W)
        if g_frames is not None and frames == T:      # g is zero outside frames [t0, t0 + n) of the (single) clip
            d.out_t0, d.out_tn = int(g_frames[0]), int(g_frames[1])
        nbytes = L.lib().dat_conv3d_wgrad_workspace_bytes(C.byref(d), self.cin, self.cout)
        wsb = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        if out is not None:
            assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == self.w.numel()
        dW = out if out is not None else torch.empty(self.w.shape, dtype=torch.float32, device=x.device)
        ctx().call('dat_conv3d_wgrad', _stream(), C.byref(d), _ptr(x), _ptr(g), self.g_cstride, self.cin, self.cout,
                   _ptr(self.scale), _ptr(wsb), _ptr(dW))
        if self.scale is None or not want_dscale:
            return dW, None
        # dL/dscale[c] = <w[c], G[c]> with G = dW / scale (the reference's AffineChannelNd has no such gradient; test hook)
        dscale = (dW * self.w).sum(dim=(1, 2, 3, 4)) / self.scale
        return dW, dscale

    def weight_acc(self, x, g, T, gt, g_frames=None):
        """Deferred-finish form of `weight` (dat_conv3d_wgrad_acc): ADDS this conv's weight gradient, unscaled and in the kernels'
        [tap][Cout][Cin] order, into the caller's fp32 accumulator `gt` (numel of the weight; zeroed by the caller once per iteration);
        WeightFinishBatch turns all accumulators into gradients with one launch.  Returns False -- nothing launched -- when the layer
        does not take the direct kernels (the caller then uses `weight`)."""
        frames, H, W, _ = x.shape
        d = self._fwd_desc(frames, T, H, W)
        if g_frames is not None and frames == T:
            d.out_t0, d.out_tn = int(g_frames[0]), int(g_frames[1])
        if not L.lib().dat_conv3d_wgrad_acc_supported(ctx().h, C.byref(d), self.g_cstride):
            return False
        assert gt.dtype == torch.float32 and gt.is_contiguous() and gt.numel() == self.w.numel()
        ctx().call('dat_conv3d_wgrad_acc', _stream(), C.byref(d), _ptr(x), _ptr(g), self.g_cstride, self.cin, self.cout, _ptr(gt))
        return True

    def weight_acc_job(self, x, g, T, gt, g_frames=None):
        """The arguments of `weight_acc` as a job for `wgrad_acc_batch` (None when the layer does not take the direct kernels).  The job holds
        references to x / g / gt: they stay alive (and must stay unmodified) until the batch has been enqueued."""
        frames, H, W, _ = x.shape
        d = self._fwd_desc(frames, T, H, W)
        if g_frames is not None and frames == T:
            d.out_t0, d.out_tn = int(g_frames[0]), int(g_frames[1])
        if not L.lib().dat_conv3d_wgrad_acc_supported(ctx().h, C.byref(d), self.g_cstride):
            return None
        assert gt.dtype == torch.float32 and gt.is_contiguous() and gt.numel() == self.w.numel()
        assert x.is_contiguous() and g.is_contiguous()
        return (d, x, g, self.g_cstride, self.cin, self.cout, gt)

    @property
    def pointwise(self):
        return self.kt == 1 and self.kh == 1 and self.kw == 1 and tuple(self.pads) == (0, 0, 0)

    def data(self, g, T, H, W, accumulate_into=None, g_frames=None, mask=None, inplace=True):
        """g [frames,Ho,Wo,g_cstride] -> dL/dx [frames,H,W,round64(Cin)] (added to `accumulate_into` when given).  mask: the conv's
        forward input x = relu(...) in the shape of dL/dx -- the ReLU backward of x's producer is applied in this conv's epilogue
        (dL/dx := x > 0 ? dL/dx : 0, res_mode 3), which saves that producer's elementwise mask pass.  mask AND accumulate_into: the sum
        with the other contribution and the mask in one epilogue (res_mode 4)."""
        if self._data_layer is None:
            # flipped / transposed / scale-folded weights are packed straight from the forward master (no ATen flip, mul, copy)
            pads = (self.kt - 1 - self.pads[0], self.kh - 1 - self.pads[1], self.kw - 1 - self.pads[2])
            self._data_layer = ConvLayer(None, None, None, stride=(1, 1), pads=pads, relu=False, dtype=self.dtype,
                                         cin_stride=self.g_cstride, dgrad_of=(self.w, self.scale))
        frames, Ho, Wo, cs = g.shape
        Hz, Wz = H - self.kh + 1 + 2 * self.pads[1], W - self.kw + 1 + 2 * self.pads[2]
        if self.stride[0] == 2 or (Hz, Wz) != (Ho, Wo):
            gz = torch.empty((frames, Hz, Wz, cs), dtype=g.dtype, device=g.device)
            if self.stride[0] == 2:
                ctx().call('dat_zero_insert2x', _stream(), self.dtype, _ptr(g), _ptr(gz), frames, Ho, Wo, Hz, Wz, cs)
            else:
                raise AssertionError('stride-1 conv whose output extent differs from H - K + 1 + 2p')
        else:
            gz = g
        lay = self._data_layer
        in_t = g_frames if (g_frames is not None and frames == T) else None   # zero frames of g: their temporal taps are skipped
        if accumulate_into is not None and mask is not None:
            # the LAST contribution to the gradient of x = relu(...): sum into `accumulate_into` and apply the ReLU backward of x's producer in the
            # same epilogue (res_mode 4: out = x > 0 ? conv + out : 0, in place)
            # (inplace=False: into a new tensor -- a queued weight-gradient job still reads `accumulate_into`)
            assert mask.is_contiguous() and mask.dtype == gz.dtype and mask.shape == accumulate_into.shape
            return lay(gz, T=T, residual=mask, res_mode=4, addend=accumulate_into, out=accumulate_into if inplace else None, in_t=in_t)
        if accumulate_into is not None:
            # inplace=False: the sum goes to a NEW tensor (`accumulate_into` is read as the residual and left untouched -- somebody else,
            # a queued weight-gradient job, still needs its present contents)
            return lay(gz, T=T, residual=accumulate_into, res_mode=1, out=accumulate_into if inplace else None, in_t=in_t)
        if mask is not None:
            assert mask.is_contiguous() and mask.dtype == gz.dtype
            return lay(gz, T=T, residual=mask, res_mode=3, in_t=in_t)
        return lay(gz, T=T, in_t=in_t)


class BucketAllReduce(object):
    """Gradient exchange through the C ABI (dat_allreduce_bucket = ncclAllReduce of RCCL, in place, on the current stream).  The
    128-byte communicator id is made on rank 0 and handed to the other ranks by `share_id` (default: a torch.distributed object
    broadcast, whatever backend that group uses); world size 1 needs no rendezvous."""

    def __init__(self, rank=0, world=1, share_id=None):
        self.rank, self.world = int(rank), int(world)
        idbuf = (C.c_char * 128)()
        if self.rank == 0:
            ctx().call('dat_comm_unique_id', C.cast(idbuf, C.c_void_p))
        raw = bytes(idbuf.raw)
        if self.world > 1:
            if share_id is None:
                import torch.distributed as dist
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
            else:
                raw = share_id(raw)
        idbuf = C.create_string_buffer(raw, 128)
        h = C.c_void_p()
        ctx().call('dat_comm_init_rank', C.cast(idbuf, C.c_void_p), self.world, self.rank, C.byref(h))
        self.h = h

    def all_reduce(self, flat, bucket_elems):
        """Sum `flat` (contiguous fp32 CUDA tensor) over the ranks, one call per bucket."""
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        n = flat.numel()
        for off in range(0, n, bucket_elems):
            cnt = min(bucket_elems, n - off)
            ctx().call('dat_allreduce_bucket', _stream(), self.h, C.c_void_p(flat.data_ptr() + 4 * off), C.c_size_t(cnt))

    def reduce_slice(self, flat, off, cnt):
        """Sum elements [off, off + cnt) of `flat` over the ranks, in place, on the CURRENT stream (one ncclAllReduce)."""
        assert flat.dtype == torch.float32 and flat.is_contiguous() and 0 <= off and off + cnt <= flat.numel()
        ctx().call('dat_allreduce_bucket', _stream(), self.h, C.c_void_p(flat.data_ptr() + 4 * off), C.c_size_t(cnt))

    def close(self):
        if self.h:
            L.lib().dat_comm_destroy(self.h)
            self.h = None


class PackBatch(object):
    """One-launch re-pack of many ConvLayers from their fp32 masters (dat_conv3d_pack_weights_batch): the table of entries lives on
    the device and is valid as long as the layers' master / packed buffers are (training: flat parameter buffer, persistent layers)."""

    def __init__(self, layers):
        assert layers and len({l.dtype for l in layers}) == 1
        self.layers = list(layers)
        self.dtype = layers[0].dtype
        # one launch per tap count: the kernel's LDS tile is sized by the largest entry of a launch, and a pointwise entry (most of a
        # bottleneck network's parameters) beside a 27-tap one would run at the 27-tap tile's two blocks per CU
        groups = {}
        for l in layers:
            groups.setdefault(l.kt * l.kh * l.kw, []).append(l)
        self.launches = []       # (device table, entries, blocks, taps)
        for ntap in sorted(groups):
            grp = groups[ntap]
            items = (L.PackItem * len(grp))()
            total = 0
            for i, l in enumerate(grp):
                d = l.desc(1, 1, 8, 8)
                n = L.lib().dat_conv3d_pack_item(ctx().h, C.byref(d), _ptr(l.w_src), l.cout_real, l.cin_real, int(l.is_dgrad),
                                                 _ptr(l.dgrad_scale) if l.is_dgrad else None, _ptr(l.packed), C.byref(items[i]))
                if n <= 0:
                    ctx().check(n if n < 0 else -1)
                assert items[i].ntap == ntap
                items[i].tile0 = total
                total += n
            raw = np.frombuffer(items, dtype=np.uint8).copy()
            self.launches.append((torch.from_numpy(raw).to(layers[0].packed.device), len(grp), total, ntap))

    def run(self):
        for table, n, total, ntap in self.launches:
            ctx().call('dat_conv3d_pack_weights_batch', _stream(), _ptr(table), n, total, ntap, self.dtype)
        for l in self.layers:
            if l.bias_src is not None:
                l.bias[:l.cout_real] = l.bias_src.float()


def roi_key(seed, it, stream, index):
    """The counter-based draw key of dat_sample_rois (csrc/labels.hip roi_key), on NumPy uint32 arrays: two murmur3 finalisers over
    seed / iteration / stream / candidate index.  Host mirror for tests and for anyone who has to reproduce a draw."""
    u = np.uint32
    with np.errstate(over='ignore'):
        def fmix(h):
            h = h ^ (h >> u(16)); h = h * u(0x85EBCA6B); h = h ^ (h >> u(13)); h = h * u(0xC2B2AE35); return h ^ (h >> u(16))
        idx = np.asarray(index, dtype=np.uint32)
        h = u(seed & 0xffffffff) ^ (idx * u(0x9E3779B9)) ^ (u(it & 0xffffffff) * u(0x85EBCA6B)) ^ (u(stream) * u(0xC2B2AE35))
        return fmix(fmix(h) ^ u((seed >> 32) & 0xffffffff))


def sample_rois(props, n_props, gt_boxes, gt_classes, gt_kps, T, num_classes, cls_agnostic, num_keypoints, heatmap_size, rois_per_im,
                fg_rois_per_im, fg_thresh, bg_hi, bg_lo, reg_weights, im_scale, seed, it, want_picked=False):
    """dat_sample_rois on CUDA tensors: props fp32 [cap, 4T+1] + device count (int32[1]), gt_boxes fp32 [G, 4T] (image scale), gt_classes
    int32 [G], gt_kps int32 [G, 3, K*T] or None.  Returns a dict of capacity-sized device tensors + `counts` (int32[6] on the device)."""
    dev = props.device
    G, C4 = int(gt_boxes.shape[0]), 4 * T
    Kc = 2 if cls_agnostic else num_classes
    d = L.RoiSampleDesc()
    d.T, d.num_classes, d.cls_agnostic, d.num_keypoints, d.heatmap_size = T, num_classes, int(bool(cls_agnostic)), int(num_keypoints), int(heatmap_size)
    d.rois_per_im, d.fg_rois_per_im = int(rois_per_im), int(fg_rois_per_im)
    d.fg_thresh, d.bg_thresh_hi, d.bg_thresh_lo = float(fg_thresh), float(bg_hi), float(bg_lo)
    for k in range(4):
        d.reg_weights[k] = float(reg_weights[k])
    d.im_scale = float(im_scale)
    d.seed_lo, d.seed_hi, d.iter = int(seed) & 0xffffffff, (int(seed) >> 32) & 0xffffffff, int(it) & 0xffffffff
    out = {'rois': torch.empty((rois_per_im, C4 + 1), dtype=torch.float32, device=dev),
           'labels_int32': torch.empty((rois_per_im,), dtype=torch.int32, device=dev),
           'bbox_targets': torch.empty((rois_per_im, C4 * Kc), dtype=torch.float32, device=dev),
           'bbox_inside_weights': torch.empty((rois_per_im, C4 * Kc), dtype=torch.float32, device=dev),
           'bbox_outside_weights': torch.empty((rois_per_im, C4 * Kc), dtype=torch.float32, device=dev),
           'counts': torch.zeros((8,), dtype=torch.int32, device=dev)}
    if gt_kps is not None:
        KK = num_keypoints * T
        assert gt_kps.dtype == torch.int32 and tuple(gt_kps.shape) == (G, 3, KK) and gt_kps.is_contiguous()
        out['keypoint_rois'] = torch.empty((fg_rois_per_im, C4 + 1), dtype=torch.float32, device=dev)
        out['keypoint_locations_int32'] = torch.empty((fg_rois_per_im, KK), dtype=torch.int32, device=dev)
        out['keypoint_weights'] = torch.empty((fg_rois_per_im, KK), dtype=torch.float32, device=dev)
    picked = torch.full((rois_per_im + fg_rois_per_im,), -1, dtype=torch.int32, device=dev) if want_picked else None
    assert props.dtype == torch.float32 and props.is_contiguous() and props.shape[1] == C4 + 1 and n_props.dtype == torch.int32
    assert gt_boxes.dtype == torch.float32 and gt_boxes.is_contiguous() and gt_classes.dtype == torch.int32
    ctx().call('dat_sample_rois', _stream(), C.byref(d), _ptr(props), _ptr(n_props), int(props.shape[0]), _ptr(gt_boxes), _ptr(gt_classes),
               _ptr(gt_kps), G, _ptr(out['rois']), _ptr(out['labels_int32']), _ptr(out['bbox_targets']), _ptr(out['bbox_inside_weights']),
               _ptr(out['bbox_outside_weights']), _ptr(out.get('keypoint_rois')), _ptr(out.get('keypoint_locations_int32')),
               _ptr(out.get('keypoint_weights')), _ptr(out['counts']), _ptr(picked))
    if want_picked:
        out['picked'] = picked
    return out


def wgrad_acc_batch(jobs):
    """dat_conv3d_wgrad_acc_batch: the deferred-finish weight gradients of `jobs` (ConvGrad.weight_acc_job tuples) in one call -- the
    pointwise layers among them as grouped launches that share the CUs (a tenth of the float-atomic traffic of one launch per layer)."""
    if not jobs:
        return
    arr = (L.WgradJob * len(jobs))()
    for i, (d, x, g, g_cs, cin, cout, gt) in enumerate(jobs):
        arr[i].desc = C.pointer(d)
        arr[i].x, arr[i].g, arr[i].Gt = x.data_ptr(), g.data_ptr(), gt.data_ptr()
        arr[i].g_cstride, arr[i].Cin_real, arr[i].Cout_real = g_cs, cin, cout
    ctx().call('dat_conv3d_wgrad_acc_batch', _stream(), arr, len(jobs))


class WeightFinishBatch(object):
    """One launch for the finish of many deferred weight gradients (dat_wgrad_finish_batch): entries = (gt accumulator, scale or None,
    dW fp32 [Cout, Cin, KT, KH, KW] view, accumulate).  The device table is valid as long as those buffers are (training: flat buffers)."""

    def __init__(self, entries):
        assert entries
        items = (L.WFinishItem * len(entries))()
        total = 0
        for i, (gt, scale, dW, acc) in enumerate(entries):
            cout, cin = int(dW.shape[0]), int(dW.shape[1])
            ntaps = dW.numel() // (cout * cin)
            assert gt.numel() == dW.numel() and dW.is_contiguous() and gt.is_contiguous()
            items[i].Gt, items[i].dW = gt.data_ptr(), dW.data_ptr()
            items[i].scale = scale.data_ptr() if scale is not None else None
            items[i].Cout, items[i].Cin, items[i].ntaps, items[i].accumulate = cout, cin, ntaps, int(bool(acc))
            items[i].block0 = total
            total += (dW.numel() + 2047) // 2048
        self.n, self.total = len(entries), total
        self.keep = [e[:3] for e in entries]
        raw = np.frombuffer(items, dtype=np.uint8).copy()
        self.table = torch.from_numpy(raw).to(entries[0][0].device)

    def run(self):
        ctx().call('dat_wgrad_finish_batch', _stream(), _ptr(self.table), self.n, C.c_longlong(self.total))


def _convgrad_repack(self):
    """after an in-place update of the forward master: refresh the packed data-gradient weights (if they were built)"""
    if self._data_layer is not None:
        self._data_layer.repack()


ConvGrad.repack = _convgrad_repack


def relu_bias_bwd(dy, y, dtype, C_real, relu=True, dy2=None, dbias=None):
    """g = (dy [+ dy2]) * (y > 0); dbias[c] += sum over positions.  NDHWC tensors [.., cs]."""
    cs = dy.shape[-1]
    npos = dy.numel() // cs
    if not relu and dy2 is None and dbias is not None and C_real == cs and dy.is_contiguous():
        # nothing to mask, no padded channels to clear: only the bias reduction runs (one read), the gradient tensor is dy itself
        ctx().call('dat_relu_bias_bwd', _stream(), dtype, _ptr(dy), None, None, None, _ptr(dbias), C.c_longlong(npos), C_real, cs, 0)
        return dy
    g = torch.empty_like(dy)
    ctx().call('dat_relu_bias_bwd', _stream(), dtype, _ptr(dy), _ptr(dy2), _ptr(y), _ptr(g), _ptr(dbias),
               C.c_longlong(npos), C_real, cs, int(relu))
    return g


def upsample2x_bwd(g, dtype, dtop=None):
    frames, H2, W2, cs = g.shape
    acc = dtop is not None
    if dtop is None:
        dtop = torch.empty((frames, H2 // 2, W2 // 2, cs), dtype=g.dtype, device=g.device)
    ctx().call('dat_upsample2x_bwd', _stream(), dtype, _ptr(g), _ptr(dtop), frames, H2 // 2, W2 // 2, cs, int(acc))
    return dtop


def sgd_momentum(w, v, grad, lr, momentum, weight_decay, is_bias):
    assert w.dtype == v.dtype == grad.dtype == torch.float32 and w.numel() == v.numel() == grad.numel()
    ctx().call('dat_sgd_momentum', _stream(), _ptr(w), _ptr(v), _ptr(grad), C.c_longlong(w.numel()), C.c_float(lr),
               C.c_float(momentum), C.c_float(weight_decay), int(is_bias))


def roi_align_bwd(dfeats, scales, dtype, rois, dout, T, Tr, t0, pooled, sampling, k_min=2, canon_scale=224., canon_level=4):
    """dfeats: fp32 CUDA gradient maps [frames,H,W,C] per level (finest first), accumulated in place."""
    nl = len(dfeats)
    ptrs = (C.c_void_p * nl)(*[f.data_ptr() for f in dfeats])
    Hs = (C.c_int * nl)(*[int(f.shape[1]) for f in dfeats])
    Ws = (C.c_int * nl)(*[int(f.shape[2]) for f in dfeats])
    sc = (C.c_float * nl)(*[float(v) for v in scales])
    Cc = int(dfeats[0].shape[3])
    assert all(f.dtype == torch.float32 and f.shape[3] == Cc for f in dfeats) and dout.shape[-1] == Cc
    R = rois.shape[0]
    ctx().call('dat_roi_align_bwd', _stream(), dtype, ptrs, Hs, Ws, sc, nl, k_min, C.c_float(canon_scale), canon_level, T, Cc,
               _ptr(rois), R, Tr, t0, pooled, sampling, _ptr(dout))


def kps_finalize_bwd(dout, dtype, R, Tr, S, cs, K, up):
    dsub = torch.empty((R * Tr, S, S, cs), dtype=tdtype(dtype), device=dout.device)
    ctx().call('dat_kps_finalize_bwd', _stream(), dtype, _ptr(dout), R, Tr, S, cs, K, up, _ptr(dsub))
    return dsub


def rpn_loss(head, dtype, A, logit_off, delta_off, labels_wide, targets_wide, inside_wide, outside_wide, cls_mult, beta,
             bbox_mult, loss2, T=1, per_frame=False):
    """head [N (*T if per_frame),H,W,cs]; label arrays in the reference layouts (CUDA); returns dhead.  loss2: fp32 CUDA [2]
    accumulated."""
    F_, H, W, cs = head.shape
    N = F_ // T if per_frame else F_
    Hw, Ww = int(labels_wide.shape[2]), int(labels_wide.shape[3])
    dhead = torch.empty_like(head)
    ctx().call('dat_rpn_loss', _stream(), dtype, _ptr(head), _ptr(dhead), N, H, W, cs, A, T, int(per_frame), logit_off, delta_off,
               _ptr(labels_wide), _ptr(targets_wide), _ptr(inside_wide), _ptr(outside_wide), Hw, Ww, C.c_float(cls_mult),
               C.c_float(beta), C.c_float(bbox_mult), _ptr(loss2))
    return dhead


def smooth_l1_rows(pred, dtype, D, targets, inside, outside, beta, mult, loss):
    """pred [R, ld] (any leading shape, last dim = ld) -> dpred same shape/dtype."""
    ld = pred.shape[-1]
    R = pred.numel() // ld
    dpred = torch.empty_like(pred)
    ctx().call('dat_smooth_l1_rows', _stream(), dtype, _ptr(pred), ld, _ptr(targets), _ptr(inside), _ptr(outside), R, D,
               C.c_float(beta), C.c_float(mult), _ptr(dpred), _ptr(loss))
    return dpred


def softmax_ce_rows(logits, dtype, D, labels, weights, mult, loss, correct=None, out_dtype=None):
    ld = logits.shape[-1]
    R = logits.numel() // ld
    odt = dtype if out_dtype is None else out_dtype
    dl = torch.empty(logits.shape, dtype=tdtype(odt), device=logits.device)
    ctx().call('dat_softmax_ce_rows', _stream(), dtype, _ptr(logits), ld, _ptr(labels), _ptr(weights), R, D, C.c_float(mult),
               odt, _ptr(dl), ld, _ptr(loss), _ptr(correct))
    return dl


def anchor_overlaps(anchors, gts, T, im_h, im_w, straddle, out=None):
    """Device half of the RPN anchor labelling (dat_anchor_overlaps): anchors [n,4T], gts [G,4T] fp32 CUDA ->
    (a2g_max [n] fp32 (-1 = outside the image), a2g_arg [n] int32, best [n] uint8)."""
    n, G = anchors.shape[0], gts.shape[0]
    assert anchors.dtype == gts.dtype == torch.float32 and anchors.is_contiguous() and gts.is_contiguous()
    assert anchors.shape[1] == 4 * T and (G == 0 or gts.shape[1] == 4 * T)
    if out is None:
        out = (torch.empty(n, dtype=torch.float32, device=anchors.device), torch.empty(n, dtype=torch.int32, device=anchors.device),
               torch.empty(n, dtype=torch.uint8, device=anchors.device))
    gmax = torch.empty(max(G, 1), dtype=torch.int32, device=anchors.device)
    ctx().call('dat_anchor_overlaps', _stream(), _ptr(anchors), n, _ptr(gts) if G else None, G, T, C.c_float(im_h), C.c_float(im_w),
               C.c_float(straddle), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(gmax))
    return out


def scatter_words(dst, offsets, values):
    """dst.view(32-bit words)[offsets[i]] = values[i] (dat_scatter_words); dst any 4-byte dtype, offsets int32, values 4-byte."""
    assert dst.element_size() == 4 and values.element_size() == 4 and offsets.dtype == torch.int32 and dst.is_contiguous()
    ctx().call('dat_scatter_words', _stream(), _ptr(dst), C.c_longlong(dst.numel()), _ptr(offsets), _ptr(values), offsets.numel())


def stem_pack(data, dtype):
    """data fp32 [N,3,T,H,W] -> packed [N*T, Ho+3, Wo, 64] (see dat_hip.h)."""
    data = data.contiguous()
    n, c, t, h, w = data.shape
    assert c == 3
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    out = torch.empty((n * t, ho + 3, wo, 64), dtype=tdtype(dtype), device=data.device)
    ctx().call('dat_stem_pack', _stream(), _ptr(data), _ptr(out), dtype, n, t, h, w)
    return out


def stem_layer(conv1_w, scale, bias, dtype):
    """conv1 [Cout,3,1,7,7] -> ConvLayer over the packed stem input (KH=4, KW=1, pad 0)."""
    cout = conv1_w.shape[0]
    w4 = torch.empty((cout, 64, 1, 4, 1), dtype=torch.float32, device=conv1_w.device)
    ctx().call('dat_stem_weights', _stream(), _ptr(conv1_w.contiguous().float()), cout, _ptr(w4))
    return ConvLayer(w4, scale, bias, stride=(1, 1), pads=(0, 0, 0), relu=True, dtype=dtype, cin_stride=64)


class StemConv(object):
    """conv1 [64,3,1,7,7] + AffineChannelNd + ReLU on the NC(T)HW fp32 `data` blob in ONE kernel (dat_stem_conv)."""

    def __init__(self, conv1_w, scale, bias, dtype, relu=True):
        w = conv1_w.contiguous().float()
        assert tuple(w.shape[1:]) == (3, 1, 7, 7) and w.shape[0] == 64, w.shape
        self.dtype, self.relu = dtype, bool(relu)
        self.scale = None if scale is None else scale.contiguous().float()
        self.bias = None if bias is None else bias.contiguous().float()
        self.packed = torch.empty(L.lib().dat_stem_conv_weight_bytes(dtype), dtype=torch.uint8, device=w.device)
        ctx().call('dat_stem_conv_pack_weights', _stream(), dtype, _ptr(w), 64, _ptr(self.packed))

    def __call__(self, data):
        data = data.contiguous()
        n, c, t, h, w = data.shape
        assert c == 3 and data.dtype == torch.float32
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out = torch.empty((n * t, ho, wo, 64), dtype=tdtype(self.dtype), device=data.device)
        ctx().call('dat_stem_conv', _stream(), self.dtype, _ptr(data), _ptr(self.packed), _ptr(self.scale), _ptr(self.bias),
                   int(self.relu), n, t, h, w, _ptr(out))
        return out

    def pooled(self, data):
        """conv1 + affine + ReLU + MaxPool [1,3,3]/[1,2,2]/pad 1 in one kernel (dat_stem_conv_pool): returns pool1 only."""
        data = data.contiguous()
        n, c, t, h, w = data.shape
        assert c == 3 and data.dtype == torch.float32
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        hp, wp = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
        out = torch.empty((n * t, hp, wp, 64), dtype=tdtype(self.dtype), device=data.device)
        ctx().call('dat_stem_conv_pool', _stream(), self.dtype, _ptr(data), _ptr(self.packed), _ptr(self.scale), _ptr(self.bias),
                   int(self.relu), n, t, h, w, _ptr(out))
        return out


def _stem_pooled_u8(self, fb):
    """conv1 + affine + ReLU + pool1 straight from uploaded uint8 frames (dat_stem_conv_pool_u8): `fb` = utils.blob.FrameBlob -- the `data`
    blob that was never materialised.  Bit-identical to `pooled(fb.materialise())`."""
    frames = fb.frames
    assert frames.dtype == torch.uint8 and frames.is_contiguous() and frames.dim() == 4 and frames.shape[3] == 3
    F, h, w, _ = [int(v) for v in frames.shape]
    (oh, ow), (ph, pw) = fb.out_hw, fb.pad_hw
    ho, wo = (ph - 1) // 2 + 1, (pw - 1) // 2 + 1
    hp, wp = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
    out = torch.empty((F, hp, wp, 64), dtype=tdtype(self.dtype), device=frames.device)
    means = (C.c_double * 3)(*[float(v) for v in np.asarray(fb.pixel_means, dtype=np.float64).reshape(-1)[:3]])
    ctx().call('dat_stem_conv_pool_u8', _stream(), self.dtype, _ptr(frames), F, h, w, C.c_double(fb.scale), C.c_double(fb.scale), oh, ow, ph, pw,
               means, _ptr(self.packed), _ptr(self.scale), _ptr(self.bias), int(self.relu), _ptr(out))
    return out


StemConv.pooled_u8 = _stem_pooled_u8


def maxpool_hw(x, dtype, k, stride, pad):
    f, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = torch.empty((f, ho, wo, c), dtype=x.dtype, device=x.device)
    ctx().call('dat_maxpool_hw', _stream(), dtype, _ptr(x), _ptr(out), f, h, w, c, k, stride, pad)
    return out


def time_avg(x, dtype, n, t):
    f, h, w, c = x.shape
    assert f == n * t
    out = torch.empty((n, h, w, c), dtype=x.dtype, device=x.device)
    ctx().call('dat_time_avg', _stream(), dtype, _ptr(x), _ptr(out), n, t, C.c_longlong(h * w * c))
    return out


def roi_align(feats, scales, dtype, rois, T, Tr, t0, pooled, sampling, k_min=2, canon_scale=224., canon_level=4):
    """feats: list of [N*T,H,W,C] (finest level FIRST, i.e. level k_min first); rois fp32 [R, 4Tr+1] (CUDA)."""
    nl = len(feats)
    lv = (L.RoiLevel * nl)()
    for i, (f, s) in enumerate(zip(feats, scales)):
        lv[i].feat = f.data_ptr()
        lv[i].H, lv[i].W = f.shape[1], f.shape[2]
        lv[i].spatial_scale = s
    c = feats[0].shape[3]
    r = rois.shape[0]
    out = torch.empty((r * Tr, pooled, pooled, c), dtype=feats[0].dtype, device=feats[0].device)
    ctx().call('dat_roi_align', _stream(), dtype, lv, nl, k_min, C.c_float(canon_scale), canon_level, T, c,
               _ptr(rois), r, Tr, t0, pooled, sampling, _ptr(out))
    return out


def spatial_mean(x, dtype, c):
    f, h, w, cs = x.shape
    out = torch.empty((f, c), dtype=torch.float32, device=x.device)
    ctx().call('dat_spatial_mean', _stream(), dtype, _ptr(x), _ptr(out), f, h * w, c, cs)
    return out


def softmax_rows(x, k):
    rows, ld = x.shape
    out = torch.empty((rows, k), dtype=torch.float32, device=x.device)
    ctx().call('dat_softmax_rows', _stream(), _ptr(x), _ptr(out), rows, k, ld, k)
    return out


class RpnLevelSpec(object):
    def __init__(self, head, H, W, A, T, feat_stride, cstride, logit_off, delta_off, frame, anchors,
                 apply_sigmoid=True, per_frame=False):
        self.head, self.H, self.W, self.A, self.T = head, H, W, A, T
        self.feat_stride, self.cstride = feat_stride, cstride
        self.logit_off, self.delta_off, self.frame = logit_off, delta_off, frame
        self.anchors = anchors  # fp32 CUDA [A, 4T]
        self.apply_sigmoid = apply_sigmoid
        self.per_frame = per_frame


def zeros_packed(dev, specs):
    """Several zero-initialised 4-byte-element tensors out of ONE zeroed allocation (one fill launch instead of one per tensor):
    specs = [(shape, torch.float32 | torch.int32), ...]; every view starts on a 16-byte boundary."""
    sizes = [(int(np.prod(shp)) + 3) // 4 * 4 for shp, _ in specs]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    ctx().call('dat_fill_zero', _stream(), _ptr(flat), C.c_size_t(flat.numel() * 4))      # (hipMemsetAsync: no ATen fill kernel in a forward)
    out, off = [], 0
    for (shp, dt), n in zip(specs, sizes):
        v = flat[off:off + int(np.prod(shp))]
        out.append((v if dt == torch.float32 else v.view(dt)).view(tuple(shp)))
        off += n
    return out


def rpn_proposals(levels, dtype, im_info, pre_nms, post_nms, nms_thresh, min_size, batch_idx=0., n_images=1, frame_stride=0):
    """One image: returns (rois [nl, post_nms, 4T+1], probs [nl, post_nms], counts int32 [nl]) CUDA tensors.  n_images > 1
    (dat_rpn_proposals_batch): image i reads frame `level.frame + i * frame_stride` of every head tensor and is clipped with
    im_info[i]; returns rois [n_images, nl, post_nms, 4T+1] (col 0 = batch_idx + i), probs [n_images, nl, post_nms], counts
    [n_images, nl] -- per image exactly what the one-image call returns."""
    nl = len(levels)
    T = levels[0].T
    dev = levels[0].head.device
    heads = (C.c_void_p * nl)(*[l.head.data_ptr() for l in levels])
    anchors = (C.c_void_p * nl)(*[l.anchors.data_ptr() for l in levels])
    lv = (L.RpnLevel * nl)()
    for i, l in enumerate(levels):
        lv[i].H, lv[i].W, lv[i].A, lv[i].T = l.H, l.W, l.A, l.T
        lv[i].feat_stride = l.feat_stride
        lv[i].cstride, lv[i].logit_off, lv[i].delta_off, lv[i].frame = l.cstride, l.logit_off, l.delta_off, l.frame
        lv[i].apply_sigmoid = int(l.apply_sigmoid)
        lv[i].per_frame = int(l.per_frame)
    im_info = np.asarray(im_info, dtype=np.float32).reshape(-1, 3)
    assert im_info.shape[0] == n_images, (im_info.shape, n_images)
    info = (C.c_float * (3 * n_images))(*[float(v) for v in im_info.reshape(-1)])
    lead = (n_images, nl) if n_images > 1 else (nl,)
    rois, probs, counts = zeros_packed(dev, [(lead + (post_nms, 4 * T + 1), torch.float32), (lead + (post_nms,), torch.float32),
                                            (lead, torch.int32)])
    ctx().call('dat_rpn_proposals_batch', _stream(), dtype, heads, lv, anchors, nl, int(n_images), int(frame_stride), info, pre_nms,
               post_nms, C.c_float(nms_thresh), C.c_float(min_size), C.c_float(batch_idx), _ptr(rois), _ptr(probs), _ptr(counts))
    return rois, probs, counts


def collect_rois(rois, probs, counts, post_nms):
    """rois [nl, cap, cols] -> (out [post_nms, cols], n_out int32[1]); batched rois [n_images, nl, cap, cols] ->
    (out [n_images * post_nms, cols] with image i in rows [i * post_nms, ...), n_out int32[n_images]): PER IMAGE top post_nms."""
    if rois.dim() == 4:
        ni, nl, cap, cols = rois.shape
    else:
        ni, (nl, cap, cols) = 1, rois.shape
    out, n_out = zeros_packed(rois.device, [((ni * post_nms, cols), torch.float32), ((ni,), torch.int32)])
    ctx().call('dat_collect_rois_batch', _stream(), _ptr(rois), _ptr(probs), _ptr(counts), nl, ni, cap, cols, post_nms,
               _ptr(out), _ptr(n_out))
    return out, n_out


def nms(dets, thresh):
    """dets CUDA fp32 [n, 4T+1]; returns CUDA int32 keep (reference ordering conventions)."""
    n, cols = dets.shape
    T = (cols - 1) // 4
    keep = torch.empty((max(n, 1),), dtype=torch.int32, device=dets.device)
    nk = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    ctx().call('dat_nms', _stream(), _ptr(dets.contiguous()), n, T, C.c_float(thresh), _ptr(keep), _ptr(nk))
    return keep[:int(nk.item())]


def nms_host(dets_np, thresh):
    """numpy in / numpy out, `_nms` convention of lib/nms/gpu_nms.hpp:3-9 (dets pre-sorted by score)."""
    dets_np = np.ascontiguousarray(dets_np, dtype=np.float32)
    n, dim = dets_np.shape
    keep = np.zeros((max(n, 1),), dtype=np.int32)
    num = C.c_int(0)
    ctx().call('dat_nms_host', keep.ctypes.data_as(C.POINTER(C.c_int)), C.byref(num),
               dets_np.ctypes.data_as(C.POINTER(C.c_float)), n, dim, C.c_float(thresh))
    return keep[:num.value]


def box_results(rois, n_rois, cls_prob, bbox_pred, num_classes, T, im_scale, im_shape, reg_weights, xform_clip, score_thresh,
                nms_thresh, detections_per_im, out_cap, cls_agnostic=False, n_images=1):
    """dat_box_results (lib/core/test.py:215-252, 750-806, 78-123 on the device).  rois CUDA fp32 [cap, 4T+1] with the DEVICE count
    n_rois (int32[1]); cls_prob [R, K], bbox_pred [R, K*4T] CUDA fp32.  Returns (dets [out_cap, 4T+2], keypoint_rois [out_cap, 4T+1],
    n_out int32[2]) on the device -- no host synchronisation.  n_images > 1 (dat_box_results_batch): rois [n_images * cap, ...] with
    counts n_rois[n_images], im_scale / im_shape sequences per image; returns dets [n_images * out_cap, 4T+2], keypoint_rois
    [n_images * out_cap, 4T+1] (col 0 = image index) and n_out int32[n_images, 2]."""
    ni = int(n_images)
    cap = int(rois.shape[0]) // ni
    assert cap * ni == int(rois.shape[0])
    scales = [float(v) for v in np.asarray(im_scale, dtype=np.float64).reshape(-1)]
    shapes = [tuple(im_shape)] * ni if not isinstance(im_shape[0], (tuple, list, np.ndarray)) else [tuple(sh) for sh in im_shape]
    if len(scales) == 1:
        scales = scales * ni
    assert len(scales) == ni and len(shapes) == ni
    ds = (L.DetDesc * ni)()
    for i in range(ni):
        d = ds[i]
        d.num_classes, d.T, d.cls_agnostic_bbox_reg, d.detections_per_im = int(num_classes), int(T), int(bool(cls_agnostic)), int(detections_per_im)
        d.im_scale, d.im_scale_f64 = scales[i], scales[i]
        d.im_h, d.im_w = int(shapes[i][0]), int(shapes[i][1])
        for k in range(4):
            d.reg_weights[k] = float(reg_weights[k])
        d.xform_clip, d.score_thresh, d.nms_thresh = float(xform_clip), float(score_thresh), float(nms_thresh)
    cols = 4 * T + 1
    wsb = torch.empty(ni * L.lib().dat_box_results_workspace_bytes(cap, int(num_classes), int(T)), dtype=torch.uint8, device=rois.device)
    dets = torch.empty((ni * out_cap, cols + 1), dtype=torch.float32, device=rois.device)
    kp = torch.empty((ni * out_cap, cols), dtype=torch.float32, device=rois.device)
    n_out = torch.empty((2,) if ni == 1 else (ni, 2), dtype=torch.int32, device=rois.device)
    assert rois.dtype == cls_prob.dtype == bbox_pred.dtype == torch.float32 and rois.is_contiguous()
    assert n_rois.numel() == ni
    ctx().call('dat_box_results_batch', _stream(), _ptr(rois), _ptr(n_rois), cap, _ptr(cls_prob), int(cls_prob.stride(0)), _ptr(bbox_pred),
               int(bbox_pred.stride(0)), ds, ni, _ptr(wsb), int(out_cap), _ptr(dets), _ptr(kp), _ptr(n_out))
    return dets, kp, n_out


def deconv_k4s2_as_conv3x3(w):
    """ConvTranspose weight fp32 [Cin, K, 4, 4] -> conv weight fp32 [4K, Cin, 1, 3, 3]."""
    cin, k = int(w.shape[0]), int(w.shape[1])
    out = torch.empty((4 * k, cin, 1, 3, 3), dtype=torch.float32, device=w.device)
    ctx().call('dat_deconv_k4s2_weights', _stream(), _ptr(w.contiguous().float()), cin, k, _ptr(out))
    return out


def kps_finalize(sub, dtype, R, Tr, K, up):
    f, s, s2, cs = sub.shape
    assert f == R * Tr and s == s2
    m = 2 * s * up
    out = torch.empty((R, Tr * K, m, m), dtype=torch.float32, device=sub.device)
    ctx().call('dat_kps_finalize', _stream(), dtype, _ptr(sub), R, Tr, s, cs, K, up, _ptr(out))
    return out


def preprocess_frames(frames, T, scale, pixel_means, pad_stride=0, out=None):
    """frames: CUDA uint8 [F, h, w, 3] (BGR, HWC) -> the `data` blob fp32 [F / T, 3, T, H, W] (dat_preprocess_frames: mean
    subtraction, cv2.INTER_LINEAR resize by `scale`, zero padding to a multiple of pad_stride), bit-identical to
    utils.blob.prep_im_for_blob + im_list_to_blob.  Returns (data, (out_h, out_w))."""
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3 and frames.is_contiguous()
    F, h, w, _ = [int(v) for v in frames.shape]
    oh, ow = int(np.rint(h * scale)), int(np.rint(w * scale))          # cvRound(src * fx): half to even
    ph, pw = (oh, ow) if not pad_stride else (int(np.ceil(oh / float(pad_stride)) * pad_stride), int(np.ceil(ow / float(pad_stride)) * pad_stride))
    if out is None:
        out = torch.empty((F // T, 3, T, ph, pw), dtype=torch.float32, device=frames.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == F * 3 * ph * pw
    means = (C.c_double * 3)(*[float(v) for v in np.asarray(pixel_means, dtype=np.float64).reshape(-1)[:3]])
    ctx().call('dat_preprocess_frames', _stream(), _ptr(frames), F, int(T), h, w, C.c_double(scale), C.c_double(scale), oh, ow, ph, pw,
               means, _ptr(out))
    return out, (oh, ow)


def heatmaps_to_keypoints(maps, boxes, T, K, min_size=0):
    """maps fp32 CUDA [R, T*K, M, M], boxes fp32 CUDA [R, 4T] -> fp32 CUDA [R, 4, T*K] rows (x, y, logit, prob)."""
    R, TK, M, M2 = maps.shape
    # boxes: [R, 4T], or wider rows whose first 4T columns are the boxes (the detection rows of dat_box_results: read in place, no slice copy)
    assert TK == T * K and M == M2 and boxes.dim() == 2 and boxes.shape[0] == R and boxes.shape[1] >= 4 * T
    assert maps.dtype == torch.float32 and boxes.dtype == torch.float32 and maps.is_contiguous() and boxes.is_contiguous()
    out = torch.empty((R, 4, TK), dtype=torch.float32, device=maps.device)
    ctx().call('dat_heatmaps_to_keypoints_ld', _stream(), _ptr(maps), _ptr(boxes), int(boxes.shape[1]), R, T, K, M, int(min_size), _ptr(out))
    return out


class ConvProfiler(object):
    """HIP-event timing of every conv launch (bench.py roofline leg)."""

    def __init__(self, capacity=4096):
        self.capacity = capacity

    def start(self):
        L.lib().dat_prof_enable(ctx().h, self.capacity)

    def stop(self):
        tags = (C.c_int * self.capacity)()
        flops = (C.c_double * self.capacity)()
        ms = (C.c_float * self.capacity)()
        n = L.lib().dat_prof_read(ctx().h, self.capacity, tags, flops, ms)
        mhz = C.c_double(0.0)
        L.lib().dat_prof_clock(ctx().h, C.byref(mhz))
        self.shader_mhz = mhz.value      # average shader clock of the conv kernel over the profiled launches
        L.lib().dat_prof_enable(ctx().h, 0)
        return [(tags[i], flops[i], ms[i]) for i in range(n)]
