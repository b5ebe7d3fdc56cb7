"""ctypes binding of libdat_hip.so (C ABI in include/dat_hip.h).

This is the ONLY way the Python host side reaches the kernels: plain pointers and sizes,
no torch types in the signatures.  torch is used by callers purely as a device-memory /
stream provider (`tensor.data_ptr()`, `torch.cuda.current_stream().cuda_stream`).

There is deliberately NO fallback: if the shared library is missing, importing this
module raises (build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C detectandtrack_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DAT_H16=fp16 selects the IEEE-half build of the same sources (cfg.HIP.DTYPE 'fp16'; one 16-bit flavour per process)
H16 = os.environ.get('DAT_H16', 'bf16')
assert H16 in ('bf16', 'fp16'), 'DAT_H16: bf16 | fp16'
LIB_PATH = os.environ.get('DAT_LIB', os.path.join(_HERE, 'libdat_hip_f16.so' if H16 == 'fp16' else 'libdat_hip.so'))

DAT_F32, DAT_BF16, DAT_BF16X3 = 0, 1, 2
DAT_OK = 0


class DatError(RuntimeError):
    """Raised for a non-zero C-ABI return code; message = dat_last_error() (the CAFFE_ENFORCE analogue)."""


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        'dtype', 'frames', 'T', 'H', 'W', 'Cin', 'Cout', 'out_cstride', 'KT', 'KH', 'KW',
        'stride_h', 'stride_w', 'pad_t', 'pad_h', 'pad_w', 'relu', 'res_mode', 'out_t0', 'out_tn', 'in_t0', 'in_tn')]


class PackItem(C.Structure):
    _fields_ = [('w', C.c_void_p), ('packed', C.c_void_p), ('scale', C.c_void_p)] + \
               [(n, C.c_int) for n in ('rows', 'cols', 'ntap', 'cout_pad', 'cin', 'frag', 'dgrad', 'dtype', 'tile0', 'tiles_x', 'cit')]


class WFinishItem(C.Structure):
    _fields_ = [('Gt', C.c_void_p), ('scale', C.c_void_p), ('dW', C.c_void_p)] + \
               [(n, C.c_int) for n in ('Cout', 'Cin', 'ntaps', 'accumulate')] + [('block0', C.c_longlong)]


class WgradJob(C.Structure):
    _fields_ = [('desc', C.POINTER(ConvDesc)), ('x', C.c_void_p), ('g', C.c_void_p), ('g_cstride', C.c_int), ('Cin_real', C.c_int),
                ('Cout_real', C.c_int), ('Gt', C.c_void_p)]


class RoiSampleDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('T', 'num_classes', 'cls_agnostic', 'num_keypoints', 'heatmap_size', 'rois_per_im', 'fg_rois_per_im')] + \
               [(n, C.c_float) for n in ('fg_thresh', 'bg_thresh_hi', 'bg_thresh_lo')] + [('reg_weights', C.c_float * 4), ('im_scale', C.c_float)] + \
               [(n, C.c_uint) for n in ('seed_lo', 'seed_hi', 'iter')]


class RoiLevel(C.Structure):
    _fields_ = [('feat', C.c_void_p), ('H', C.c_int), ('W', C.c_int), ('spatial_scale', C.c_float)]


class DetDesc(C.Structure):
    _fields_ = [('num_classes', C.c_int), ('T', C.c_int), ('cls_agnostic_bbox_reg', C.c_int), ('detections_per_im', C.c_int),
                ('im_scale', C.c_float), ('im_scale_f64', C.c_double), ('im_h', C.c_int), ('im_w', C.c_int),
                ('reg_weights', C.c_float * 4), ('xform_clip', C.c_float), ('score_thresh', C.c_float), ('nms_thresh', C.c_float)]


class RpnLevel(C.Structure):
    _fields_ = [('H', C.c_int), ('W', C.c_int), ('A', C.c_int), ('T', C.c_int), ('feat_stride', C.c_float),
                ('cstride', C.c_int), ('logit_off', C.c_int), ('delta_off', C.c_int), ('frame', C.c_int),
                ('apply_sigmoid', C.c_int), ('per_frame', C.c_int)]


if not os.path.exists(LIB_PATH):
    raise ImportError('libdat_hip.so not built at %s — run __graft_entry__.build()' % LIB_PATH)
_lib = C.CDLL(LIB_PATH)

_p, _i, _f, _ll, _d = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_double
_PROTOS = {
    'dat_version': (_i, []),
    'dat_h16_format': (_i, []),
    'dat_ctx_create': (_i, [C.POINTER(_p), _i]),
    'dat_ctx_destroy': (None, [_p]),
    'dat_last_error': (C.c_char_p, [_p]),
    'dat_ws_info': (_i, [_p, C.POINTER(_p), C.POINTER(C.c_size_t), C.POINTER(_i)]),
    'dat_ws_reserve': (_i, [_p, C.c_size_t]),
    'dat_fill_zero': (_i, [_p, _p, _p, C.c_size_t]),
    'dat_prof_enable': (_i, [_p, _i]),
    'dat_prof_read': (_i, [_p, _i, C.POINTER(_i), C.POINTER(_d), C.POINTER(_f)]),
    'dat_prof_clock': (_i, [_p, C.POINTER(_d)]),
    'dat_zero_even_fwd': (_i, [_p, _p, _p, _ll]),
    'dat_affine_channel_nd_fwd': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _ll]),
    'dat_affine_channel_nd_bwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _ll]),
    'dat_ncdhw_to_ndhwc': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i]),
    'dat_ndhwc_to_ncdhw': (_i, [_p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i]),
    'dat_conv3d_out_shape': (_i, [C.POINTER(ConvDesc), C.POINTER(_i), C.POINTER(_i)]),
    'dat_conv3d_packed_weight_bytes': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'dat_conv3d_pack_weights': (_i, [_p, _p, C.POINTER(ConvDesc), _p, _i, _i, _p]),
    'dat_conv3d_pack_weights_dgrad': (_i, [_p, _p, C.POINTER(ConvDesc), _p, _i, _i, _p, _p]),
    'dat_comm_unique_id': (_i, [_p, _p]),
    'dat_comm_init_rank': (_i, [_p, _p, _i, _i, C.POINTER(C.c_void_p)]),
    'dat_allreduce_bucket': (_i, [_p, _p, _p, _p, C.c_size_t]),
    'dat_comm_destroy': (_i, [_p]),
    'dat_conv3d_pack_item': (_i, [_p, C.POINTER(ConvDesc), _p, _i, _i, _i, _p, _p, C.POINTER(PackItem)]),
    'dat_conv3d_pack_weights_batch': (_i, [_p, _p, _p, _i, _i, _i, _i]),
    'dat_conv3d_fwd': (_i, [_p, _p, C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p]),
    'dat_conv3d_fwd_x3': (_i, [_p, _p, C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p]),
    'dat_conv3d_fwd_sum_mask': (_i, [_p, _p, C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p]),
    'dat_conv3d_tune_plan': (_i, [_p, _i, _i]),
    'dat_conv3d_persistent_share': (_i, [_p, _i]),
    'dat_conv3d_flops': (_d, [C.POINTER(ConvDesc), _i, _i]),
    'dat_stem_pack': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i]),
    'dat_stem_weights': (_i, [_p, _p, _p, _i, _p]),
    'dat_split_bf16x2': (_i, [_p, _p, _p, _p, _ll, _i]),
    'dat_maxpool_hw': (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i]),
    'dat_time_avg': (_i, [_p, _p, _i, _p, _p, _i, _i, _ll]),
    'dat_copy_frames': (_i, [_p, _p, _p, C.POINTER(_i), _p, C.POINTER(_i), _i, _ll]),
    'dat_roi_align': (_i, [_p, _p, _i, C.POINTER(RoiLevel), _i, _i, _f, _i, _i, _i, _p, _i, _i, _i, _i, _i, _p]),
    'dat_spatial_mean': (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i]),
    'dat_softmax_rows': (_i, [_p, _p, _p, _p, _i, _i, _i, _i]),
    'dat_rpn_proposals': (_i, [_p, _p, _i, C.POINTER(_p), C.POINTER(RpnLevel), C.POINTER(_p), _i, C.POINTER(_f),
                               _i, _i, _f, _f, _f, _p, _p, _p]),
    'dat_collect_rois': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    'dat_rpn_proposals_batch': (_i, [_p, _p, _i, C.POINTER(_p), C.POINTER(RpnLevel), C.POINTER(_p), _i, _i, _i, C.POINTER(_f),
                                     _i, _i, _f, _f, _f, _p, _p, _p]),
    'dat_collect_rois_batch': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    'dat_nms': (_i, [_p, _p, _p, _i, _i, _f, _p, _p]),
    'dat_nms_host': (_i, [_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), _i, _i, _f]),
    '_nms': (None, [C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), _i, _i, _f, _i]),
    'dat_box_results_workspace_bytes': (C.c_size_t, [_i, _i, _i]),
    'dat_box_results': (_i, [_p, _p, _p, _p, _i, _p, _i, _p, _i, C.POINTER(DetDesc), _p, _i, _p, _p, _p]),
    'dat_box_results_batch': (_i, [_p, _p, _p, _p, _i, _p, _i, _p, _i, C.POINTER(DetDesc), _i, _p, _i, _p, _p, _p]),
    'dat_soft_nms_host': (_i, [C.POINTER(_f), _i, _f, _f, _f, _i, C.POINTER(_f), C.POINTER(_i), C.POINTER(_i)]),
    'dat_deconv_k4s2_weights': (_i, [_p, _p, _p, _i, _i, _p]),
    'dat_kps_finalize': (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    'dat_stem_conv_weight_bytes': (C.c_size_t, [_i]),
    'dat_stem_conv_pack_weights': (_i, [_p, _p, _i, _p, _i, _p]),
    'dat_stem_conv': (_i, [_p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'dat_stem_conv_pool': (_i, [_p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'dat_preprocess_frames': (_i, [_p, _p, _p, _i, _i, _i, _i, _d, _d, _i, _i, _i, _i, C.POINTER(_d), _p]),
    'dat_stem_conv_pool_u8': (_i, [_p, _p, _i, _p, _i, _i, _i, _d, _d, _i, _i, _i, _i, C.POINTER(_d), _p, _p, _p, _i, _p]),
    'dat_heatmaps_to_keypoints': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'dat_heatmaps_to_keypoints_ld': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'dat_conv3d_wgrad_workspace_bytes': (C.c_size_t, [C.POINTER(ConvDesc), _i, _i]),
    'dat_conv3d_wgrad': (_i, [_p, _p, C.POINTER(ConvDesc), _p, _p, _i, _i, _i, _p, _p, _p]),
    'dat_conv3d_wgrad_acc_supported': (_i, [_p, C.POINTER(ConvDesc), _i]),
    'dat_conv3d_wgrad_acc': (_i, [_p, _p, C.POINTER(ConvDesc), _p, _p, _i, _i, _i, _p]),
    'dat_wgrad_finish_batch': (_i, [_p, _p, _p, _i, C.c_longlong]),
    'dat_conv3d_wgrad_acc_batch': (_i, [_p, _p, C.POINTER(WgradJob), _i]),
    'dat_relu_bias_bwd': (_i, [_p, _p, _i, _p, _p, _p, _p, _p, C.c_longlong, _i, _i, _i]),
    'dat_zero_insert2x': (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i]),
    'dat_upsample2x_bwd': (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i]),
    'dat_sgd_momentum': (_i, [_p, _p, _p, _p, _p, C.c_longlong, _f, _f, _f, _i]),
    'dat_roi_align_bwd': (_i, [_p, _p, _i, C.POINTER(_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), _i, _i, _f, _i, _i, _i,
                               _p, _i, _i, _i, _i, _i, _p]),
    'dat_kps_finalize_bwd': (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    'dat_rpn_loss': (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _i, _f, _f, _f, _p]),
    'dat_smooth_l1_rows': (_i, [_p, _p, _i, _p, _i, _p, _p, _p, _i, _i, _f, _f, _p, _p]),
    'dat_anchor_overlaps': (_i, [_p, _p, _p, _i, _p, _i, _i, _f, _f, _f, _p, _p, _p, _p]),
    'dat_scatter_words': (_i, [_p, _p, _p, C.c_longlong, _p, _p, _i]),
    'dat_sample_rois': (_i, [_p, _p, C.POINTER(RoiSampleDesc), _p, _p, _i, _p, _p, _p, _i] + [_p] * 10),
    'dat_softmax_ce_rows': (_i, [_p, _p, _i, _p, _i, _p, _p, _i, _i, _f, _i, _p, _i, _p, _p]),
}
EXPORTS = sorted(_PROTOS)
for _name, (_res, _args) in _PROTOS.items():
    _fn = getattr(_lib, _name)  # AttributeError here == header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


class Ctx(object):
    """One context per GPU process (reference: one process per GPU at inference, lib/utils/subprocess.py:38-63)."""

    def __init__(self, device=0):
        h = _p()
        rc = _lib.dat_ctx_create(C.byref(h), int(device))
        if rc != DAT_OK:
            raise DatError('dat_ctx_create(device=%d) failed with %d (no visible MI355X?)' % (device, rc))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, 'h', None):
            _lib.dat_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != DAT_OK:
            raise DatError('%s (code %d)' % (_lib.dat_last_error(self.h).decode(), rc))

    def call(self, name, *args):
        self.check(getattr(_lib, name)(self.h, *args))


def lib():
    return _lib
