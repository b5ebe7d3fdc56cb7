/* libdat_hip.so — C ABI of the MI355X-native DetectAndTrack hot path.
 *
 * Plain C: opaque context, device pointers, sizes, POD descriptors. No C++ or torch
 * types cross this boundary; errors are return codes + dat_last_error() text (the
 * replacement for Caffe2's CAFFE_ENFORCE message, cf. reference
 * tests/test_zero_even_op.py:43).  Unless a function says "host", every pointer is a
 * DEVICE pointer owned by the caller; nothing here synchronises the stream.
 *
 * What each entry point replaces in the reference (facebookresearch/DetectAndTrack):
 *   - the Caffe2 operator-plugin boundary of lib/ops (REGISTER_CUDA_OPERATOR,
 *     lib/ops/affine_channel_nd_op.cu:95-98; loaded via lib/utils/c2.py:53-56), and
 *   - the one real C ABI in the tree, lib/nms/gpu_nms.hpp:3-9 (`_nms`), and
 *   - the external Caffe2/cuDNN ops called by name from the lib/modeling builders
 *     (ConvNd, MaxPool, RoIAlign, ConvTranspose, FC, ...; census in SURVEY.md §3.5).
 *
 * Internal activation layout: NDHWC ("frames x H x W x C", C contiguous, C % 64 == 0),
 * fp32 (parity mode) or bf16 (performance mode).  The reference's NC[T]HW fp32 blobs
 * appear only at the boundary (dat_ncdhw_to_ndhwc / dat_ndhwc_to_ncdhw, dat_stem_pack,
 * dat_kps_finalize).
 */
#ifndef DAT_HIP_H_
#define DAT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dat_ctx dat_ctx;
typedef void* dat_stream; /* hipStream_t */

enum { DAT_F32 = 0, DAT_BF16 = 1,
       /* conv descriptors only: fp32 activations in HBM, conv arithmetic on hi / lo bf16 splits of both operands
        * (x_hi*W_hi + x_hi*W_lo + x_lo*W_hi, fp32 accumulate; ~2^-16 relative).  dat_conv3d_fwd then takes x = the split
        * tensor made by dat_split_bf16x2 (pixel pitch 2 * Cin bf16), w_packed = dat_conv3d_pack_weights of a DAT_BF16 descriptor
        * with Cin' = 3 * Cin over the master [W_hi | W_lo | W_hi] per 64-channel chunk; y / residual are fp32. */
       DAT_BF16X3 = 2 };
enum { DAT_OK = 0, DAT_ERR_ARG = -1, DAT_ERR_LAUNCH = -2, DAT_ERR_ALLOC = -3, DAT_ERR_UNSUPPORTED = -4 };

/* ---- context ------------------------------------------------------------------------- */
int dat_version(void);
/* The 16-bit element format of THIS build of the library: 0 = bfloat16 (libdat_hip.so, the benched performance mode), 1 = IEEE half
 * (libdat_hip_f16.so, the same sources compiled with -DDAT_H16_IS_FP16: every tensor or packed weight tagged DAT_BF16 then holds fp16 and the
 * conv kernels issue v_mfma_f32_32x32x16_f16 -- the bf16 MFMA rate with three more mantissa bits; DAT_BF16X3 is not available there). */
int dat_h16_format(void);
int dat_ctx_create(dat_ctx** out, int device);
void dat_ctx_destroy(dat_ctx* ctx);
const char* dat_last_error(dat_ctx* ctx);
/* The context's scratch buffer (proposal path, split-K partials, bias partials): current device pointer, size and
 * the number of times it has grown.  Growth never frees the outgrown buffer before dat_ctx_destroy, so launches
 * captured into a hipGraph (which have the old address baked in) stay replayable.  Any out pointer may be NULL. */
int dat_ws_info(dat_ctx* ctx, void** ptr, size_t* bytes, int* generation);
/* Make sure the scratch holds at least `bytes` (tests / callers that want the growth outside a timed region). */
int dat_ws_reserve(dat_ctx* ctx, size_t bytes);
/* hipMemsetAsync(ptr, 0, bytes) on the stream (capturable): zero-initialised kernel outputs without a framework fill launch */
int dat_fill_zero(dat_ctx* ctx, dat_stream s, void* ptr, size_t bytes);

/* Per-launch HIP-event timing of the conv kernel (used by bench.py's roofline leg).
 * enable: start recording (capacity launches); read: sync the events and return
 * n records {tag, flops, milliseconds}; tag = BN*10000 + BP*10 + dtype. */
int dat_prof_enable(dat_ctx* ctx, int capacity);
int dat_prof_read(dat_ctx* ctx, int max_records, int* tags, double* flops, float* ms);
/* Average shader clock (MHz) the conv kernel ran at since dat_prof_enable: every block adds its s_memtime (shader
 * cycles) and s_memrealtime (100 MHz) deltas; MI355X drops well below its 2.4 GHz peak clock under MFMA load. */
int dat_prof_clock(dat_ctx* ctx, double* shader_mhz);

/* ---- toy op: ZeroEven  (lib/ops/zero_even_op.cu:25-56) --------------------------------- */
int dat_zero_even_fwd(dat_ctx* ctx, dat_stream s, float* x, long long n);

/* ---- AffineChannelNd  (lib/ops/affine_channel_nd_op.cu:20-92) -------------------------- */
/* NC(spatial...) fp32, y = x*scale[c] + bias[c]; in-place allowed (.cc:23-24). */
int dat_affine_channel_nd_fwd(dat_ctx* ctx, dat_stream s, const float* x, const float* scale, const float* bias,
                              float* y, int N, int C, long long inner);
/* gradient: dX = dY*scale[c]; no dscale/dbias (affine_channel_nd_op.cc:29-37). */
int dat_affine_channel_nd_bwd(dat_ctx* ctx, dat_stream s, const float* dy, const float* scale, float* dx, int N,
                              int C, long long inner);

/* ---- boundary layout moves --------------------------------------------------------------- */
/* src NC(T)HW fp32 [N,C,T,H,W] -> dst [N*T,H,W,Cs] (channels >= C zero-filled). */
int dat_ncdhw_to_ndhwc(dat_ctx* ctx, dat_stream s, const float* src, void* dst, int dtype, int N, int C, int T,
                       int H, int W, int Cs);
/* src [N*T,H,W,Cs] -> dst NC(T)HW fp32 [N,C,T,H,W] (first C channels). */
int dat_ndhwc_to_ncdhw(dat_ctx* ctx, dat_stream s, const void* src, int dtype, float* dst, int N, int C, int T,
                       int H, int W, int Cs);

/* ---- fused 3D convolution: ConvNd/Conv/FC + bias|AffineChannelNd + Sum + Relu ------------ */
/* Replaces ConvNd (lib/modeling/ResNet3D.py:258, detector.py:421-433), AffineChannelNd
 * (a3), Relu, Sum (ResNet3D.py:146-154), UpsampleNearest+Sum (FPN3D.py:207-222), Conv
 * (FPN.py:222-262), FC (head_builder.py:34-37), ConvTranspose-as-subpixel-conv. */
typedef struct {
    int dtype;               /* DAT_F32 | DAT_BF16 (activations and packed weights) */
    int frames, T;           /* frames = N*T input frames; temporal taps never cross a multiple of T */
    int H, W, Cin;           /* input NDHWC dims; Cin = channel stride, multiple of 64 */
    int Cout;                /* outputs written per position (multiple of 4) */
    int out_cstride;         /* channel stride of y (>= Cout) */
    int KT, KH, KW;          /* kernel */
    int stride_h, stride_w;  /* temporal stride is 1 (VIDEO.TIME_STRIDE_ON unsupported: FPN3D.py:199-203) */
    int pad_t, pad_h, pad_w; /* symmetric explicit pads, Caffe2 `pads=2*[..]` */
    int relu;                /* fused Relu */
    int res_mode;            /* 0 none | 1 residual same shape | 2 residual at (h/2, w/2) (nearest 2x) | 3 MASK: y = residual > 0 ? v : 0
                                (ReLU backward fused into a data-gradient conv: residual = the forward input of the conv, same shape as y)
                                | 4 SUM + MASK: y = mask > 0 ? v + addend : 0 (dat_conv3d_fwd_sum_mask only) */
    int out_t0, out_tn;      /* out_tn > 0: write only output frames t in [out_t0, out_t0+out_tn) of every clip, stored
                                compactly as [N*out_tn, Ho, Wo, C] (the frames a following SliceKeyFrame keeps,
                                FPN3D.py:170-183 'slice-center'); out_tn == 0: all T frames.  residual (if any) is
                                indexed like the output */
    int in_t0, in_tn;        /* in_tn > 0: input frames outside [in_t0, in_t0+in_tn) of every clip are known to be ZERO (a key-frame
                                gradient embedded in its temporal window): the temporal taps that would read them are skipped */
} dat_conv_desc;

int dat_conv3d_out_shape(const dat_conv_desc* d, int* Ho, int* Wo);
/* bytes of the packed weight buffer [KT*KH*KW][Cout_pad][Cin] in d->dtype */
size_t dat_conv3d_packed_weight_bytes(const dat_conv_desc* d);
/* w: fp32 [Cout_real, Cin_real, KT, KH, KW] (reference blob layout); rows/cols beyond are zero. */
int dat_conv3d_pack_weights(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const float* w, int Cout_real,
                            int Cin_real, void* packed);
/* The same packed layout for the DATA-GRADIENT conv of a forward layer (training): d describes the
 * data-gradient conv (Cin = channel stride of the output gradient, Cout = forward input channels, same kernel, stride 1, pads
 * k-1-p); w_fwd is the forward master fp32 [CoutF, CinF, KT, KH, KW]; packs W'[ci][co][taps flipped] = w_fwd[co][ci][..] *
 * scale_fwd[co] (scale_fwd: the layer's fused AffineChannelNd scale, or NULL) in one pass -- no flipped / transposed copy. */
int dat_conv3d_pack_weights_dgrad(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const float* w_fwd, int CoutF, int CinF,
                                  const float* scale_fwd, void* packed);
/* Batched re-pack (training: every trainable layer and its data-gradient twin after each SGD step, model_builder.py:953-985 updates the
 * blobs the next forward reads): dat_conv3d_pack_item fills one table entry with the arguments of dat_conv3d_pack_weights (dgrad 0:
 * rows = Cout_real, cols = Cin_real) or dat_conv3d_pack_weights_dgrad (dgrad 1: rows = CinF, cols = CoutF, scale = scale_fwd) and returns
 * the number of thread blocks the entry takes (< 0: error); the caller sets tile0 to the running sum, copies the table to the device
 * and packs all entries (one dtype) with ONE launch of dat_conv3d_pack_weights_batch. */
typedef struct dat_pack_item {
    const float* w;
    void* packed;
    const float* scale;
    int rows, cols, ntap, cout_pad, cin, frag, dgrad, dtype;
    int tile0, tiles_x;
    int cit;                 /* input channels per block tile (set by dat_conv3d_pack_item: 64 / 32 / 16 by tap count) */
} dat_pack_item;
int dat_conv3d_pack_item(dat_ctx* ctx, const dat_conv_desc* d, const float* w, int rows_real, int cols_real, int dgrad, const float* scale,
                         void* packed, dat_pack_item* item);
int dat_conv3d_pack_weights_batch(dat_ctx* ctx, dat_stream s, const dat_pack_item* items_dev, int n, int total_blocks, int max_ntap,
                                  int dtype);
/* y = act( conv(x, w)*scale[c] + bias[c] + residual ); scale may be NULL (=1), bias may be NULL (=0).
 * scale/bias: fp32 [Cout].  residual: same dtype/stride as y. */
int dat_conv3d_fwd(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x, const void* w_packed,
                   const float* scale, const float* bias, const void* residual, void* y);
/* res_mode 4 (training; replaces the Sum of the two gradient contributions of a residual block's output + the ReLU gradient of that
 * output, i.e. what Caffe2's AddGradientOperators emits for lib/modeling/ResNet3D.py:21-87 -- a SumOp and a ReluGradient -- around the
 * ConvGradient of the block's first conv):  y = mask > 0 ? act-free(conv(x, w)*scale + bias) + addend : 0.
 * addend, mask: same dtype / shape / channel stride as y; addend may BE y (in place).  bf16 / fp32. */
int dat_conv3d_fwd_sum_mask(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x, const void* w_packed,
                            const float* scale, const float* bias, const void* addend, const void* mask, void* y);
/* bf16x3 mode (d->dtype == DAT_BF16X3) with the operand split fused into the producer: the same launch also writes the
 * hi / lo bf16 split of y (dat_split_bf16x2's layout, pixel pitch 2 * out_cstride) into y_split, bit-identical to
 * dat_split_bf16x2(y) -- the next conv reads it without a split pre-pass.  y_split may be NULL (then == dat_conv3d_fwd);
 * needs Cout == out_cstride, a multiple of 64. */
int dat_conv3d_fwd_x3(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x_split, const void* w_packed,
                      const float* scale, const float* bias, const void* residual, void* y, void* y_split);
/* algorithmic FLOPs of one launch: 2*Cout*Cin*KT*KH*KW*frames*Ho*Wo */
double dat_conv3d_flops(const dat_conv_desc* d, int Cin_real, int Cout_real);

/* ---- stem: conv1 [1,7,7]/s[1,2,2] input packing (ResNet3D.py:258-262) ---------------------- */
/* data NC(T)HW fp32 [N,3,T,H,W] -> packed [N*T, Ho+3, Wo, 64]:
 *   packed[f, r, ow, dkh*32 + kw*3 + c] = data[n, c, t, 2r-3+dkh, 2ow-3+kw] (0 outside), so that
 *   conv1 == a (KH=4, KW=1, stride 1, pad 0) conv over r with 64 channels. */
int dat_stem_pack(dat_ctx* ctx, dat_stream s, const float* data, void* packed, int dtype, int N, int T, int H,
                  int W);
/* conv1_w fp32 [64,3,1,7,7] -> fp32 [64,64,1,4,1] in the packed-channel order above (then dat_conv3d_pack_weights) */
int dat_stem_weights(dat_ctx* ctx, dat_stream s, const float* conv1_w, int Cout, float* w_k4);

/* ---- operand split of the bf16x3 conv mode (dtype DAT_BF16X3 of dat_conv_desc) ------------------ */
/* x fp32 [npos, C] (C % 64 == 0) -> y bf16 [npos, 2C]: per 64-channel chunk q, line 2q = bf16(x), line 2q + 1 =
 * bf16(x - float(bf16(x))).  HBM-bound: 4 bytes read + 4 written per element. */
int dat_split_bf16x2(dat_ctx* ctx, dat_stream s, const float* x, void* y, long long npos, int C);

/* ---- MaxPool [1,k,k]/s[1,st,st] (ResNet3D.py:263-265; FPN3D.py:158-163 with k=1) ------------- */
int dat_maxpool_hw(dat_ctx* ctx, dat_stream s, int dtype, const void* x, void* y, int frames, int H, int W, int C,
                   int k, int stride, int pad);

/* dst frame dst_idx[i] = src frame src_idx[i], i < n: frames are contiguous slabs of frame_bytes (a multiple of 16) in NDHWC.  HOST index
 * arrays (they travel in the kernel arguments).  The per-frame trunk cache of the pipelined engine (cfg.HIP.FRAME_TRUNK_CACHE; reference
 * sliding-window inference lib/utils/video.py:149-201 recomputes conv1 ... res2 for T - 1 of T frames of every clip): scatter of the new
 * frames' trunk outputs into their cache slots, gather of a forward's clips from the cache. */
int dat_copy_frames(dat_ctx* ctx, dat_stream s, const void* src, const int* src_idx, void* dst, const int* dst_idx, int n,
                    long long frame_bytes);
/* ---- time pooling (detector.py:559-576): avg over T -> frames/T frames ----------------------- */
int dat_time_avg(dat_ctx* ctx, dat_stream s, int dtype, const void* x, void* y, int N, int T, long long hwc);

/* ---- RoIAlign, tube- and FPN-level-aware (detector.py:216-310 + ops/roi_blob_transforms.py:25-36) ---- */
typedef struct {
    const void* feat;    /* [N*T, H, W, C] */
    int H, W;
    float spatial_scale;
} dat_roi_level;
/* rois fp32 [R, 4*Tr+1] (col0 = batch idx n).  Tube slot t reads frame n*T + t (Tr == T) or n*T + t0 (Tr == 1:
 * 2D heads on a key frame, detector.py:571-576).  With n_levels > 1 each RoI picks level
 * clip(floor(canon_level + log2(sqrt(mean_t area)/canon_scale + 1e-6)), k_min, k_min+n_levels-1)
 * (lib/modeling/FPN.py:349-360) and levels[lvl - k_min]; output rows stay in RoI order, which is what the
 * reference obtains with Concat + BatchPermutation(rois_idx_restore) (detector.py:283-296).
 * out [R*Tr, P, P, C].  Legacy Detectron RoIAlign: no half-pixel shift, roi size clamped to >= 1,
 * sampling_ratio^2 samples per bin (ceil(roi/P) when sampling_ratio <= 0). */
int dat_roi_align(dat_ctx* ctx, dat_stream s, int dtype, const dat_roi_level* levels, int n_levels, int k_min,
                  float canon_scale, int canon_level, int T, int C, const float* rois, int R, int Tr, int t0,
                  int pooled, int sampling_ratio, void* out);

/* ---- network input on the device (lib/utils/blob.py:40-90, lib/utils/image.py:82-93) -------------- */
/* frames: DEVICE uint8 [n_frames, h, w, 3] (BGR, HWC: decoded video frames as they are).  data: fp32 [n_frames / T, 3, T, pad_h,
 * pad_w] = im_list_to_blob(prep_im_for_blob(frame)): float32(frame - pixel_means) resized by (fx, fy) with cv2.INTER_LINEAR
 * semantics to out_h x out_w (= rint(h * fy) x rint(w * fx), computed by the caller like the reference does), zero-padded to
 * pad_h x pad_w (a multiple of FPN.COARSEST_STRIDE), channels-first, frames of a clip along T (T = 1, n_frames = N: the 2D
 * blob [N, 3, pad_h, pad_w]).  pixel_means: HOST double[3] (cfg.PIXEL_MEANS, BGR).  Bit-identical to the host path. */
int dat_preprocess_frames(dat_ctx* ctx, dat_stream s, const unsigned char* frames, int n_frames, int T, int h, int w, double fx,
                          double fy, int out_h, int out_w, int pad_h, int pad_w, const double* pixel_means, float* data);

/* ---- small head maths ------------------------------------------------------------------------- */
/* mean over H,W of [frames,H,W,C] -> [frames,C] fp32 (ReduceBackMean x2, ResNet3D.py:318-319) */
int dat_spatial_mean(dat_ctx* ctx, dat_stream s, int dtype, const void* x, float* y, int frames, int HW, int C,
                     int Cs);
/* softmax over the first K of each row of stride ld (model_builder.py:452) */
int dat_softmax_rows(dat_ctx* ctx, dat_stream s, const float* x, float* y, int rows, int K, int ld_in, int ld_out);

/* ---- RPN proposals: GenerateProposalsOp on device (lib/ops/generate_proposals.py:40-181) -------- */
typedef struct {
    int H, W, A, T;          /* grid, anchors per cell, frames per tube */
    float feat_stride;
    int cstride;             /* channel stride of the head tensor */
    int logit_off, delta_off;/* channel offsets: logits [A], deltas [A*T*4] (anchor, frame, xywh) */
    int frame;               /* which frame of the head tensor holds this image's map */
    int apply_sigmoid;       /* 1: head holds raw logits (model_builder.py:583 Sigmoid fused here); 0: probabilities */
    int per_frame;           /* tube RPN on a head tensor that keeps its T frames: logits = mean over frames frame..frame+T-1
                                (TimePool avg, model_builder.py:532), deltas of tube slot t read from frame+t at channel
                                delta_off + a*4 (model_builder.py:545-563 regroups exactly this into (a, t, xywh)) */
} dat_rpn_level;

/* head: conv output [frames,H,W,cstride] (fp32 or bf16) holding RAW logits (sigmoid applied here,
 * model_builder.py:583) and deltas.  anchors: fp32 [A, 4T] cell anchors (generate_anchors.py).
 * Per level l writes (score order): rois_out + l*post_nms*(4T+1), probs_out + l*post_nms, count[l].
 * Semantics: top pre_nms (<= 16384 per level) by score (ties: lower (h,w,a) index first) -> bbox_transform, weights 1
 * (boxes.py:141-183) -> clip (boxes.py:243-253) -> min_size*im_scale filter AND-ed over frames
 * (generate_proposals.py:184-196) -> NMS (>= thresh boxes / > thresh tubes) -> first post_nms. */
int dat_rpn_proposals(dat_ctx* ctx, dat_stream s, int dtype, const void* const* heads, const dat_rpn_level* levels,
                      const float* const* anchors, int n_levels, const float* im_info /* HOST [3]: H, W, scale */,
                      int pre_nms, int post_nms, float nms_thresh, float min_size, float batch_idx,
                      float* rois_out, float* probs_out, int* counts_out);

/* CollectAndDistributeFpnRpnProposalsOp.collect (lib/ops/collect_and_distribute_fpn_rpn_proposals.py:44-62).
 * In: per-level rois/probs/counts as written by dat_rpn_proposals (level stride = level_cap rows of roi_cols).
 * Out: rois [n_out, roi_cols] = global top post_nms by score (ties: lower concat index), n_out (dev int32[1]).
 * The per-level split (`distribute`, :65-87) is not materialised on device: dat_roi_align assigns levels in-kernel. */
int dat_collect_rois(dat_ctx* ctx, dat_stream s, const float* rois_lvls, const float* probs_lvls, const int* counts,
                     int n_levels, int level_cap, int roi_cols, int post_nms, float* rois, int* n_out);

/* Several images per forward (round 3; the reference's GenerateProposalsOp loops `for im_i in range(num_images)`,
 * lib/ops/generate_proposals.py:133-147, and runs inference with one image per batch, lib/core/test.py:212-214).  All images share
 * the level geometry; image i reads frame `levels[l].frame + i * frame_stride` of every head tensor, is clipped / filtered with
 * im_info[3*i .. 3*i+2] (HOST, n_images x 3) and writes rois_out [n_images, n_levels, post_nms, 4T+1] (col 0 = batch_idx + i),
 * probs_out [n_images, n_levels, post_nms], counts_out [n_images, n_levels].  Every image's result is what dat_rpn_proposals
 * returns for it alone (same kernels, the image is a grid dimension).  n_images <= DAT_MAX_IMAGES. */
#define DAT_MAX_IMAGES 16
int dat_rpn_proposals_batch(dat_ctx* ctx, dat_stream s, int dtype, const void* const* heads, const dat_rpn_level* levels,
                            const float* const* anchors, int n_levels, int n_images, int frame_stride, const float* im_info,
                            int pre_nms, int post_nms, float nms_thresh, float min_size, float batch_idx, float* rois_out,
                            float* probs_out, int* counts_out);
/* collect PER IMAGE (inference keeps RPN_POST_NMS_TOP_N proposals per image, exactly what a one-image forward keeps): in
 * [n_images, n_levels, level_cap, ...] as written by dat_rpn_proposals_batch, out rois [n_images, post_nms, roi_cols] (rows past
 * n_out[i] of image i are not written), n_out int32[n_images]. */
int dat_collect_rois_batch(dat_ctx* ctx, dat_stream s, const float* rois_lvls, const float* probs_lvls, const int* counts,
                           int n_levels, int n_images, int level_cap, int roi_cols, int post_nms, float* rois, int* n_out);

/* ---- NMS (lib/utils/cython_nms.pyx:37-87, lib/nms/py_cpu_nms_tubes.py:17-53) ---------------------- */
/* dets: dev fp32 [n, 4T+1], ANY order.  keep: dev int32[n]; boxes (T==1): ascending original indices
 * (cython_nms semantics, suppress at IoU >= thresh); tubes (T>1): score order, suppress at mean IoU > thresh.
 * num_keep: dev int32[1]. */
int dat_nms(dat_ctx* ctx, dat_stream s, const float* dets, int n, int T, float thresh, int* keep, int* num_keep);
/* Host-pointer convenience wrapper with the reference's `_nms` convention (lib/nms/gpu_nms.hpp:3-9):
 * boxes_host [boxes_num, boxes_dim] PRE-SORTED by score, keep_out has room for boxes_num ints.
 * Synchronous; returns 0 or a DAT_ERR code (the reference only printed errors). */
int dat_nms_host(dat_ctx* ctx, int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                 float nms_overlap_thresh);
/* The reference's own C symbol, exact prototype (lib/nms/gpu_nms.hpp:3-9; caller lib/nms/gpu_nms.pyx:14-34 links against it
 * unchanged): host pointers, boxes [boxes_num, 5] pre-sorted by score, suppression at IoU > thresh (strict, as
 * lib/nms/nms_kernel.cu:71; the Cython CPU path and dat_nms / dat_nms_host use >=), keep_out = kept row positions in the given
 * order, synchronous, selects `device_id`, errors are printed (never returned), one internal context per device, thread-safe. */
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float nms_overlap_thresh,
          int device_id);

/* ---- detection post-processing between `model.net` and `model.keypoint_net`, on the device ----------------------------------
 * Replaces the reference's host glue (a device sync + NumPy per clip): lib/core/test.py:215-252 (rois / im_scale, bbox_transform
 * with MODEL.BBOX_REG_WEIGHTS, clip to the image), :750-806 box_results_with_nms_and_limit (per class: score > SCORE_THRESH,
 * NMS at TEST.NMS; then TEST.DETECTIONS_PER_IM over all classes: scores >= the D-th best) and :78-123 _get_rois_blob for the
 * keypoint net.  Soft-NMS / box voting are not covered here (host path, lib/utils/cython_nms.pyx:98-203). */
typedef struct {
    int num_classes;            /* K incl. background (class 0 is skipped) */
    int T;                      /* frames per tube (1 = boxes) */
    int cls_agnostic_bbox_reg;  /* MODEL.CLS_AGNOSTIC_BBOX_REG: bbox_pred holds one 4T group (its last columns) */
    int detections_per_im;      /* TEST.DETECTIONS_PER_IM (0 = no limit) */
    float im_scale;             /* im_info scale (float32 copy, informational) */
    double im_scale_f64;        /* the float64 scale NumPy divides / multiplies by (test.py:216, :123) */
    int im_h, im_w;             /* shape of the UNSCALED image: clip bounds (test.py:229) */
    float reg_weights[4];       /* MODEL.BBOX_REG_WEIGHTS (wx, wy, ww, wh) */
    float xform_clip;           /* BBOX_XFORM_CLIP = log(1000/16) */
    float score_thresh;         /* TEST.SCORE_THRESH */
    float nms_thresh;           /* TEST.NMS */
} dat_det_desc;
size_t dat_box_results_workspace_bytes(int roi_cap, int num_classes, int T);
/* rois fp32 [roi_cap, 4T+1] with *n_rois valid rows (DEVICE count), cls_prob fp32 [R, prob_ld >= K], bbox_pred fp32
 * [R, pred_ld >= K*4T] (class-major, then frame, then xyxy).  Outputs, class-major then NMS output order (= np.vstack of the
 * reference's cls_boxes): dets_out fp32 [out_cap, 4T+2] rows (box, score, class); keypoint_rois fp32 [out_cap, 4T+1] rows
 * (0, box * im_scale), zero rows past the kept ones; n_out int32[2] = {rows written (<= out_cap), rows the limit rule keeps}: when
 * n_out[1] > out_cap (exact score ties at the D-th place) the caller must take the host path to honour the reference. */
int dat_box_results(dat_ctx* ctx, dat_stream s, const float* rois, const int* n_rois, int roi_cap, const float* cls_prob, int prob_ld,
                    const float* bbox_pred, int pred_ld, const dat_det_desc* d, void* workspace, int out_cap, float* dets_out,
                    float* keypoint_rois, int* n_out);
/* The same for n_images images of one forward (each image's rows are what dat_box_results returns for it alone): image i owns rois
 * rows [i*roi_cap, (i+1)*roi_cap) with n_rois[i] valid ones (and the same rows of cls_prob / bbox_pred), is decoded with d[i]
 * (im_scale / im_h / im_w per image; classes, T, weights and thresholds from d[0]) and writes dets_out [n_images, out_cap, 4T+2],
 * keypoint_rois [n_images, out_cap, 4T+1] (col 0 = i: the image's index in the batch, what RoIAlign reads as the batch index;
 * zero rows -- col 0 included -- past the kept ones) and n_out int32[n_images, 2].  workspace: n_images x
 * dat_box_results_workspace_bytes. */
int dat_box_results_batch(dat_ctx* ctx, dat_stream s, const float* rois, const int* n_rois, int roi_cap, const float* cls_prob,
                          int prob_ld, const float* bbox_pred, int pred_ld, const dat_det_desc* d, int n_images, void* workspace,
                          int out_cap, float* dets_out, float* keypoint_rois, int* n_out);

/* Soft-NMS, HOST pointers, host arithmetic (lib/utils/cython_nms.pyx:98-203 statement for statement in C float): boxes_in
 * [n, 5] -> boxes_out [*n_out <= n, 5] (re-scored, in the order the greedy loop leaves them) and inds_out (rows of boxes_in).
 * method 0 = hard, 1 = linear, 2 = gaussian (nms_wrapper.py:37).  Buffers hold n rows.  Off in every shipped config. */
int dat_soft_nms_host(const float* boxes_in, int n, float sigma, float Nt, float threshold, int method, float* boxes_out,
                      int* inds_out, int* n_out);

/* ---- keypoint head tail: ConvTranspose k4s2p1 (as 3x3 sub-pixel conv) + bilinear up (detector.py:348-380) -- */
/* Expand kps_score_lowres_w fp32 [Cin, K, 4, 4] (Caffe2 ConvTranspose layout) into an equivalent 3x3 conv
 * weight fp32 [4*K, Cin, 1, 3, 3] whose output channel (a*2+b)*K + k is sub-pixel (a,b) of keypoint k. */
int dat_deconv_k4s2_weights(dat_ctx* ctx, dat_stream s, const float* w, int Cin, int K, float* w3x3);
/* sub [R*Tr, S, S, cs] (4K sub-pixel channels, bias already added) -> kps_score fp32 NCHW
 * [R, Tr*K, 2*up*S, 2*up*S]: pixel-shuffle to 2S then the fixed bilinear ConvTranspose (k=2*up, s=up,
 * p=up/2; model_builder.py:858-868 incl. time->channel ordering t*K+k). */
int dat_kps_finalize(dat_ctx* ctx, dat_stream s, int dtype, const void* sub, int R, int Tr, int S, int cs, int K,
                     int up, float* out);

/* Tuning hook (not a reference interface): override the launch plan of the dat_conv3d_fwd calls that follow ON THIS
 * CONTEXT -- positions per block 128 | 256 and the split-K factor 1..8; 0 = back to the built-in makespan model.
 * Per-context state (no process globals): other contexts / devices of the process are unaffected.  Used by
 * tools/tune_plan.py to check the model against measured per-layer timings. */
int dat_conv3d_tune_plan(dat_ctx* ctx, int positions_per_block, int ksplit);
/* Engine hook (not a reference interface): the share of the CUs the PERSISTENT HBM-bound conv kernels (one block per CU: res2's 3x3
 * convs, the 64 -> 256 lateral, the weights-in-LDS 1x1 kernel) launched ON THIS CONTEXT take, 1..100 percent (default 100, or
 * DAT_PERSIST_PCT).  An engine that keeps several forwards in flight on their own streams lowers it so that the other forwards'
 * MFMA-bound kernels find free CUs beside them (core/pipeline.py: cfg.HIP.PERSISTENT_CU_SHARE); results do not depend on it. */
int dat_conv3d_persistent_share(dat_ctx* ctx, int percent);

/* ---- conv1, fused (ResNet3D.py:258-262): ConvNd [1,7,7] / [1,2,2] / pad [0,3,3] on `data` fp32 [N,3,T,H,W] + AffineChannelNd
 * (scale, bias: fp32 [64] or NULL) + ReLU -> out [N*T, Ho, Wo, 64] in `dtype`; weights packed once by
 * dat_stem_conv_pack_weights (dat_stem_conv_weight_bytes bytes).  Supersedes dat_stem_pack + dat_conv3d_fwd for conv1. */
size_t dat_stem_conv_weight_bytes(int dtype);
int dat_stem_conv_pack_weights(dat_ctx* ctx, dat_stream s, int dtype, const float* conv1_w, int Cout, void* packed);
int dat_stem_conv(dat_ctx* ctx, dat_stream s, int dtype, const float* data, const void* w_packed, const float* scale,
                  const float* bias, int relu, int N, int T, int H, int W, void* out);

/* conv1 + AffineChannelNd + ReLU + pool1 fused (ResNet3D.py:258-265: ... -> MaxPool [1,3,3] / [1,2,2] / pad [0,1,1]): writes
 * only out_pool [N*T, Hp, Wp, 64]; the `conv1` blob (the largest of the network, one reader) is never materialised.
 * Bit-identical to dat_stem_conv followed by dat_maxpool_hw(k 3, stride 2, pad 1). */
int dat_stem_conv_pool(dat_ctx* ctx, dat_stream s, int dtype, const float* data, const void* w_packed, const float* scale,
                       const float* bias, int relu, int N, int T, int H, int W, void* out_pool);
/* The same launch fed from the UPLOADED uint8 frames (round 6): frames = DEVICE uint8 [n_frames, h, w, 3] (BGR, HWC), the geometry arguments
 * of dat_preprocess_frames (fx = fy = the test scale, out_h x out_w the resized image, pad_h x pad_w the blob).  The patch loader computes the
 * value the `data` blob would hold -- dat_preprocess_frames' arithmetic, bit for bit -- so pool1 is IDENTICAL to dat_preprocess_frames +
 * dat_stem_conv_pool and the fp32 blob (12 bytes per network pixel) is never written or read.  out_pool [n_frames, Hp, Wp, 64] in frame
 * order (frame n*T + t of the blob is input frame n*T + t).  The frames allocation must be readable up to the next multiple of 4 bytes. */
int dat_stem_conv_pool_u8(dat_ctx* ctx, dat_stream s, int dtype, const unsigned char* frames, int n_frames, int h, int w, double fx, double fy,
                          int out_h, int out_w, int pad_h, int pad_w, const double* pixel_means, const void* w_packed, const float* scale,
                          const float* bias, int relu, void* out_pool);

/* ---- gradient exchange of data-parallel training (lib/modeling/model_builder.py:938-942: NCCLAllreduce / muji.Allreduce over the
 * parameter gradients; losses are pre-divided by NUM_GPUS, so the reduction is a plain sum).  Thin wrappers over RCCL (loaded on first
 * use).  The host owns the rendezvous: rank 0 calls dat_comm_unique_id and ships the 128 bytes to the other ranks; every rank calls
 * dat_comm_init_rank on its own device; dat_allreduce_bucket sums `count` floats in place on stream s (one call per bucket of the flat
 * gradient buffer). ---- */
typedef struct dat_comm dat_comm;
int dat_comm_unique_id(dat_ctx* ctx, void* id128);
int dat_comm_init_rank(dat_ctx* ctx, const void* id128, int nranks, int rank, dat_comm** out);
int dat_allreduce_bucket(dat_ctx* ctx, dat_stream s, dat_comm* comm, float* buf, size_t count);
int dat_comm_destroy(dat_comm* comm);

/* ---- keypoint heatmap decoding  (lib/utils/keypoints.py:94-149 heatmaps_to_keypoints, :210-216) ---- */
/* maps fp32 [R, T*K, M, M] (kps_score), boxes fp32 [R, 4*T] image-space tubes -> out fp32 [R, 4, T*K], rows
 * (x, y, logit, prob), column t*K + k (core/test.py:875-893 concatenates the frames along the keypoint axis):
 * bicubic (OpenCV INTER_CUBIC) resize of every map to the RoI's ceil size (>= min_size when min_size > 0,
 * KRCNN.INFERENCE_MIN_SIZE), argmax cell centre mapped to the image, spatial-softmax probability of that cell. */
int dat_heatmaps_to_keypoints(dat_ctx* ctx, dat_stream s, const float* maps, const float* boxes, int R, int T, int K,
                              int M, int min_size, float* out);
/* the same with `boxes` rows box_ld floats apart (>= 4T): detection rows [4T box | score | class] are read in place */
int dat_heatmaps_to_keypoints_ld(dat_ctx* ctx, dat_stream s, const float* maps, const float* boxes, int box_ld, int R, int T, int K, int M,
                                 int min_size, float* out);

/* ---- training: backward of the fused conv and the update ------------------------------ */
/* Weight gradient of a conv described like dat_conv3d_fwd (same-T, stride 1 or 2):
 *   dW[co][ci][kt][kh][kw] = sum_p g[p][co] * x[p (+) tap][ci]      (fp32, reference blob layout, overwritten)
 * x: the conv input NDHWC (channel stride d->Cin), g: gradient w.r.t. the conv output NDHWC (channel stride
 * g_cstride); d->out_t0 / out_tn > 0 (one clip) declare that g is non-zero only in those frames, which restricts the
 * reduction to their positions; scale (fp32 [Cout] or NULL) multiplies row co: the fused AffineChannelNd scale when g is the gradient
 * w.r.t. the affine OUTPUT.  Replaces Caffe2's ConvGradient filter path reached through AddGradientOperators
 * (model_builder.py:908-952).  workspace: dat_conv3d_wgrad_workspace_bytes() bytes of device memory. */
size_t dat_conv3d_wgrad_workspace_bytes(const dat_conv_desc* d, int Cin_real, int Cout_real);
int dat_conv3d_wgrad(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x, const void* g, int g_cstride,
                     int Cin_real, int Cout_real, const float* scale, void* workspace, float* dW);
/* Deferred finish (training: one zeroing and one transposing launch per ITERATION instead of one memset + one finish per layer).
 * dat_conv3d_wgrad_acc ADDS the layer's weight gradient, in the kernels' own fp32 [tap][Cout][Cin] order and without the scale, into the
 * caller's accumulator Gt -- zeroed by the caller at the start of the iteration; several calls may add into one Gt (a weight shared by
 * several convs: the RPN conv of every FPN level).  Only layers that take the direct bf16 kernels (dat_conv3d_wgrad_acc_supported == 1;
 * anything else is a DAT_ERR_ARG).  dat_wgrad_finish_batch then writes dW[co][ci][tap] (+)= scale[co] * Gt[tap][co][ci] for a table of
 * layers in one launch: an item owns the blocks [block0, block0 + ceil(Cout * Cin * ntaps / 2048)) of the grid; total_blocks = their sum. */
typedef struct dat_wfinish_item {
    const float* Gt;
    const float* scale;      /* fp32 [Cout] or NULL */
    float* dW;
    int Cout, Cin, ntaps, accumulate;
    long long block0;
} dat_wfinish_item;
int dat_conv3d_wgrad_acc_supported(dat_ctx* ctx, const dat_conv_desc* d, int g_cstride);
int dat_conv3d_wgrad_acc(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x, const void* g, int g_cstride,
                         int Cin_real, int Cout_real, float* Gt);
int dat_wgrad_finish_batch(dat_ctx* ctx, dat_stream s, const dat_wfinish_item* items_dev, int n, long long total_blocks);
/* n deferred-finish weight gradients in ONE call -- the result of n dat_conv3d_wgrad_acc calls (HOST array of jobs; every member as in that
 * function).  The pointwise layers among them (1 x 1 x 1 convs / FC, stride 1 | 2) are executed as grouped launches: a layer split over all
 * CUs adds one partial tile per CU with float atomics (~64 MB per layer at ~1.2 TB/s), a grid shared by all the layers of a gradient bucket
 * a tenth of that.  Reference semantics: the ConvGradient ops AddGradientOperators emits, lib/modeling/model_builder.py:908-951. */
typedef struct dat_wgrad_job {
    const dat_conv_desc* desc;
    const void* x;
    const void* g;
    int g_cstride, Cin_real, Cout_real;
    float* Gt;
} dat_wgrad_job;
int dat_conv3d_wgrad_acc_batch(dat_ctx* ctx, dat_stream s, const dat_wgrad_job* jobs, int n);
/* g = (dy [+ dy2]) * (y > 0 if relu) over [npos][cstride] (channels >= C zeroed); dbias[c] += sum_p g (may be NULL).
 * Relu backward on the fused conv's output + the bias / AffineChannelNd-bias reduction
 * (affine_channel_nd_op.cu:74-92). */
int dat_relu_bias_bwd(dat_ctx* ctx, dat_stream s, int dtype, const void* dy, const void* dy2, const void* y, void* g,
                      float* dbias, long long npos, int C, int cstride, int relu);
/* dst[f, 2y, 2x, :] = src[f, y, x, :], zero elsewhere: input of a stride-2 conv's data gradient run as a stride-1 conv */
int dat_zero_insert2x(dat_ctx* ctx, dat_stream s, int dtype, const void* src, void* dst, int frames, int Hs, int Ws, int Hd,
                      int Wd, int cstride);
/* FPN top-down backward (FPN3D.py:207-222): dtop[f,y,x,:] (+)= sum of the 2x2 block of g [frames, 2Ht, 2Wt, cstride] */
int dat_upsample2x_bwd(dat_ctx* ctx, dat_stream s, int dtype, const void* g, void* dtop, int frames, int Ht, int Wt,
                       int cstride, int accumulate);
/* RoIAlign backward (same sampling as dat_roi_align): dout [R*Tr, P, P, C] is scattered with fp32 atomics into the
 * per-level fp32 gradient maps dfeat_levels[l] [frames, Hs[l], Ws[l], C] (accumulated, caller zeroes them). */
int dat_roi_align_bwd(dat_ctx* ctx, dat_stream s, int dtype, float* const* dfeat_levels, const int* Hs, const int* Ws,
                      const float* scales, int n_levels, int k_min, float canon_scale, int canon_level, int T, int C,
                      const float* rois, int R, int Tr, int t0, int pooled, int sampling_ratio, const void* dout);
/* backward of dat_kps_finalize: dout fp32 [R, Tr*K, M, M] -> dsub [R*Tr, S, S, cs] (channels >= 4K zero) */
int dat_kps_finalize_bwd(dat_ctx* ctx, dat_stream s, int dtype, const float* dout, int R, int Tr, int S, int cs, int K, int up,
                         void* dsub);
/* ---- losses: value + gradient w.r.t. the prediction in one pass (model_builder.py:481-494, 612-636, 873-889; FPN.py:282-321) */
/* RPN losses of one level on the fused head tensor [N, H, W, cstride] (logits at logit_off + a, deltas at delta_off + a*4 + c):
 * SigmoidCrossEntropyLoss on the logits (labels int32 (N, A, Hw, Ww), -1 = ignore) and SmoothL1Loss on the deltas (targets /
 * inside / outside weights fp32 (N, 4A, Hw, Ww)); the wide label arrays are narrowed to H x W by indexing (SpatialNarrowAs).
 * Tube anchors (T > 1): deltas are 4*T*A channels (a, t, xywh); per_frame = 1 reads a head that keeps its T frames (logits
 * averaged over them, frame t holds the (a, xywh) deltas of slot t -- the C4 tube RPN, model_builder.py:500-609).
 * dhead gets the gradient of (loss_cls + loss_bbox) for every channel; loss2[0] += loss_cls, loss2[1] += loss_bbox. */
int dat_rpn_loss(dat_ctx* ctx, dat_stream s, int dtype, const void* head, void* dhead, int N, int H, int W, int cstride, int A,
                 int T, int per_frame, int logit_off, int delta_off, const int* labels_wide, const float* targets_wide, const float* inside_wide,
                 const float* outside_wide, int Hw, int Ww, float cls_scale_over_norm, float bbox_beta,
                 float bbox_scale_over_n, float* loss2);
/* SmoothL1Loss on row-major predictions [R, ld] (D used columns); *loss += scale_over_n * sum(out * l(in * (pred - tgt))) */
int dat_smooth_l1_rows(dat_ctx* ctx, dat_stream s, int dtype, const void* pred, int ld, const float* targets, const float* inside,
                       const float* outside, int R, int D, float beta, float scale_over_n, void* dpred, float* loss);
/* SoftmaxWithLoss over rows [R, ld] (D classes), labels int32 [R], optional weights [R]; dlogits [R, dl_ld] in dl_dtype;
 * *loss += scale_over_norm * sum_i -w_i log p_i[label_i]; *correct += #(argmax == label) when non-NULL (Accuracy op). */
int dat_softmax_ce_rows(dat_ctx* ctx, dat_stream s, int dtype, const void* logits, int ld, const int* labels, const float* weights,
                        int R, int D, float scale_over_norm, int dl_dtype, void* dlogits, int dl_ld, float* loss, int* correct);
/* MomentumSGDUpdate + the reference's gradient pre-processing (model_builder.py:954-985): biases: grad *= 2, no decay;
 * weights: grad += weight_decay * w;  v = momentum*v + lr*grad;  w -= v.  All fp32. */
int dat_sgd_momentum(dat_ctx* ctx, dat_stream s, float* w, float* v, const float* grad, long long n, float lr, float momentum,
                     float weight_decay, int is_bias);

/* ---- training input pipeline: RPN anchor labelling, device half -------------------------------------
 * Replaces the O(anchors x gts) part of reference lib/roi_data/rpn.py:283-312: the straddle filter (:283-291), the Cython
 * IoU lib/utils/cython_bbox.pyx:16-57 averaged over the tube's frames (lib/utils/boxes.py:60-69), anchor->gt max / first
 * arg-max (:297-299), gt->anchor max (:300-303) and the "anchors that attain a gt's max" flags (:304-305).
 * anchors [n, 4T] and gts [G, 4T] fp32 on the device; a2g_max [n] is -1 for anchors outside the image (straddle < 0 keeps
 * all); a2g_arg [n]; best_flag [n] bytes; gt_max [G] workspace (float bits).  Results are bit-identical to the host code. */
int dat_anchor_overlaps(dat_ctx* ctx, dat_stream s, const float* anchors, int n, const float* gts, int G, int T, float im_h, float im_w,
                        float straddle, float* a2g_max, int* a2g_arg, unsigned char* best_flag, unsigned int* gt_max);
/* dst[offsets[i]] = values[i] for 32-bit words (sparse RPN labels -> the dense "wide" label blobs of rpn.py:343-368) */
int dat_scatter_words(dat_ctx* ctx, dat_stream s, void* dst, long long dst_words, const int* offsets, const void* values, int n);

/* GenerateProposalLabels on the device (lib/ops/generate_proposal_labels.py:24-37 -> lib/roi_data/fast_rcnn.py:109-203, json_dataset.py:423-473,
 * roi_data/keypoint_rcnn.py:32-99, utils/keypoints.py:152-207).  Candidates = the G gt boxes (image scale) followed by the first *n_props
 * proposals ([cap, 4T+1], network scale, divided by im_scale); maximum overlap with the gt boxes as the reference's Cython kernel computes it;
 * rois_per_im rois are drawn, up to fg_rois_per_im foreground (overlap >= fg_thresh) first, then background ([bg_thresh_lo, bg_thresh_hi)):
 * rois [rows, 4T+1] at network scale, labels, class-specific targets / inside / outside weights [rows, 4T * (cls_agnostic ? 2 : num_classes)].
 * With gt_kps (int32 [G, 3, num_keypoints * T]): up to fg_rois_per_im keypoint rois (foreground AND a visible keypoint of their gt inside
 * their first-frame box; none at all: the gt boxes), heatmap cell index / weight per keypoint [rows, num_keypoints * T].
 * counts (device int32[8]) = rows, n_fg, keypoint rows, |fg|, |bg|, |keypoint-fg|, labelled keypoints of the keypoint rows, 0.  picked (optional int32 [rois_per_im + fg_rois_per_im]):
 * candidate index (gts first) behind every output row.
 * The draw replaces NumPy's stream by a counter-based generator: "n of S" = the n members of S with the smallest (key, index),
 * key = f(seed, iter, stream, index) (csrc/labels.hip roi_key; stream 0: fg / bg, 1: keypoint rois), in that order. */
typedef struct dat_roi_sample_desc {
    int T, num_classes, cls_agnostic, num_keypoints, heatmap_size;
    int rois_per_im, fg_rois_per_im;
    float fg_thresh, bg_thresh_hi, bg_thresh_lo;
    float reg_weights[4];
    float im_scale;
    unsigned int seed_lo, seed_hi, iter;
} dat_roi_sample_desc;
int dat_sample_rois(dat_ctx* ctx, dat_stream s, const dat_roi_sample_desc* d, const float* props, const int* n_props, int props_cap,
                    const float* gt_boxes, const int* gt_classes, const int* gt_kps, int G, float* rois, int* labels, float* targets,
                    float* w_in, float* w_out, float* kp_rois, int* kp_loc, float* kp_w, int* counts, int* picked);

#ifdef __cplusplus
}
#endif
#endif /* DAT_HIP_H_ */
