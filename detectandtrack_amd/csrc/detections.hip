// Detection post-processing on the device (gfx950): the reference's host glue between `model.net` and `model.keypoint_net`.
//
// Replaces, without a host round trip,
//   lib/core/test.py:215-252  im_detect_bbox tail: rois / im_scale, bbox_transform with MODEL.BBOX_REG_WEIGHTS
//                             (lib/utils/boxes.py:141-202), clip_tiled_boxes to the image (:243-253);
//   lib/core/test.py:750-806  box_results_with_nms_and_limit: per class score > TEST.SCORE_THRESH, NMS (TEST.NMS),
//                             then TEST.DETECTIONS_PER_IM over all classes (scores >= the D-th best);
//   lib/core/test.py:78-123   _get_rois_blob: the kept boxes * im_scale with a leading level column = `keypoint_rois`.
// The reference fetches rois / cls_prob / bbox_pred to the host (a device sync per clip), loops over classes in NumPy, calls
// the NMS, and feeds `keypoint_rois` back.  Here the chain is: one select+decode kernel and one NMS per foreground class, one
// limit+emit kernel; the keypoint net then runs on the device-resident `keypoint_rois`.
//
// Arithmetic is fp32 in NumPy's operation order (this file is compiled with -ffp-contract=off): `rois / im_scale` is a float32
// division by float32(scale) (value-based casting of the reference's NumPy 1.14, test.py:216), `boxes * im_scale` for the
// keypoint rois is formed in float64 and rounded once (test.py:98,121); exp is evaluated in double and rounded once
// (NumPy's float32 exp is within 1 ulp of that).
#include "dat_common.h"
#include "nms_internal.h"

namespace {

constexpr int DET_MAX_T = 16;
constexpr int SEL_THREADS = 1024;

constexpr int MAX_IMAGES = DAT_MAX_IMAGES;

// One block per image (blockIdx.x): image i owns rows [i * roi_cap, (i + 1) * roi_cap) of rois / prob / pred, the count n_rois[i]
// and the scratch at + i * img_ws bytes.
struct DetParams {
    const float* rois;      // [n_images * roi_cap, 4T+1]
    const int* n_rois;      // device counts [n_images]
    const float* prob;      // [R, prob_ld]
    const float* pred;      // [R, pred_ld]
    int prob_ld, pred_ld, roi_cap, T, K, cls_agnostic;
    float wx, wy, ww, wh, xform_clip, score_thresh;
    size_t img_ws;
    double im_scale[MAX_IMAGES];
    float im_h[MAX_IMAGES], im_w[MAX_IMAGES];       // the UNSCALED image (clip bounds, test.py:229)
};

// one tube of class j for roi r: boxes.py:141-183 per frame, then clip (:243-253)
// (TM: compile-time bound of the tube length -- 1 keeps the box in registers, the general instantiation's array lives in scratch;
//  see proposals.hip decode_tube)
template <int TM>
__device__ void decode_tube(const DetParams& p, int img, int r, int j, float* out) {
    const int PT = TM == 1 ? 1 : p.T;
    const int cols = 4 * PT + 1;
    const float im_w = dat_pick(p.im_w, img), im_h = dat_pick(p.im_h, img);
    const int dcls = p.cls_agnostic ? (p.K - 1) : j;       // CLS_AGNOSTIC_BBOX_REG: the last 4T columns (test.py:224-225)
#pragma unroll
    for (int t = 0; t < (TM == 1 ? 1 : PT); ++t) {
        const float* rb = p.rois + (size_t)r * cols + 1 + 4 * t;
        // `rois[:, 1:] / im_scales[0]`: float32 / float32(scale) under the reference environment's NumPy 1.14 casting
        const float sc = (float)dat_pick(p.im_scale, img);
        const float x1 = rb[0] / sc, y1 = rb[1] / sc, x2 = rb[2] / sc, y2 = rb[3] / sc;
        const float* d = p.pred + (size_t)r * p.pred_ld + ((size_t)dcls * PT + t) * 4;
        const float w = x2 - x1 + 1.0f, h = y2 - y1 + 1.0f;
        const float cx = x1 + 0.5f * w, cy = y1 + 0.5f * h;
        const float dx = d[0] / p.wx, dy = d[1] / p.wy;
        const float dw = fminf(d[2] / p.ww, p.xform_clip), dh = fminf(d[3] / p.wh, p.xform_clip);
        const float pcx = dx * w + cx, pcy = dy * h + cy;
        const float pw = (float)exp((double)dw) * w, ph = (float)exp((double)dh) * h;
        float ox1 = pcx - 0.5f * pw, oy1 = pcy - 0.5f * ph, ox2 = pcx + 0.5f * pw, oy2 = pcy + 0.5f * ph;
        ox1 = fmaxf(fminf(ox1, im_w - 1.f), 0.f);
        oy1 = fmaxf(fminf(oy1, im_h - 1.f), 0.f);
        ox2 = fmaxf(fminf(ox2, im_w - 1.f), 0.f);
        oy2 = fmaxf(fminf(oy2, im_h - 1.f), 0.f);
        out[4 * t + 0] = ox1; out[4 * t + 1] = oy1; out[4 * t + 2] = ox2; out[4 * t + 3] = oy2;
    }
}

// block-wide exclusive scan of one value per thread (SEL_THREADS threads); returns the exclusive prefix, *total = sum
__device__ unsigned block_scan(unsigned v, unsigned* scan, unsigned* total) {
    const int tid = threadIdx.x;
    scan[tid] = v;
    __syncthreads();
    for (int off = 1; off < SEL_THREADS; off <<= 1) {
        const unsigned a = (tid >= off) ? scan[tid - off] : 0u;
        __syncthreads();
        scan[tid] += a;
        __syncthreads();
    }
    const unsigned incl = scan[tid];
    *total = scan[SEL_THREADS - 1];
    __syncthreads();
    return incl - v;
}

// ---- per class: inds = where(scores[:, j] > thresh); dets_j = [decoded boxes[inds], scores[inds]]  (test.py:759-762), in roi order
template <int TM>
__global__ __launch_bounds__(SEL_THREADS) void det_select_kernel(const DetParams p, int j, float* dets, int* n_sel) {
    __shared__ unsigned scan[SEL_THREADS];
    const int img = blockIdx.x;
    const int n = min(p.n_rois[img], p.roi_cap);
    const int PT = TM == 1 ? 1 : p.T;
    const int cols = 4 * PT + 1;
    const int per = (n + SEL_THREADS - 1) / SEL_THREADS;
    const int r0 = img * p.roi_cap;              // this image's first row
    const int lo = r0 + threadIdx.x * per, hi = min(r0 + n, lo + per);
    dets = (float*)((char*)dets + (size_t)img * p.img_ws);
    n_sel = (int*)((char*)n_sel + (size_t)img * p.img_ws);
    unsigned local = 0;
    for (int r = lo; r < hi; ++r) local += p.prob[(size_t)r * p.prob_ld + j] > p.score_thresh ? 1u : 0u;
    unsigned total;
    unsigned pos = block_scan(local, scan, &total);
    float tube[4 * TM];
    for (int r = lo; r < hi; ++r) {
        const float sc = p.prob[(size_t)r * p.prob_ld + j];
        if (sc > p.score_thresh) {
            decode_tube<TM>(p, img, r, j, tube);
            float* o = dets + (size_t)pos * cols;
#pragma unroll
            for (int c = 0; c < (TM == 1 ? 4 : 4 * PT); ++c) o[c] = tube[c];
            o[4 * PT] = sc;
            ++pos;
        }
    }
    if (threadIdx.x == 0) *n_sel = (int)total;
}

struct EmitParams {
    const float* dets;     // [K-1][cap][4T+1] selected dets per foreground class
    const int* keep;       // [K-1][cap] kept rows (NMS output order)
    const int* n_keep;     // [K-1]
    int K, T, cap, D, out_cap;
    size_t img_ws;         // image i (blockIdx.x): dets / keep / n_keep at + i * img_ws bytes, outputs at slot i
    double im_scale[MAX_IMAGES];
    float* dets_out;       // [n_images][out_cap][4T+2]: box, score, class
    float* kp_rois;        // [n_images][out_cap][4T+1]: batch index (image i), box * im_scale
    int* n_out;            // [n_images][2]: rows written (<= out_cap), rows the limit rule keeps
};

// ---- DETECTIONS_PER_IM (test.py:790-800): thresh = the D-th best kept score over all classes, keep score >= thresh; then the
// class-major stack (test.py:802) and its keypoint rois.  One block; the D-th best score comes from an exact radix select on the
// float bits (scores are probabilities > 0: the bit pattern is monotonic).
__global__ __launch_bounds__(SEL_THREADS) void det_limit_emit_kernel(const EmitParams pin) {
    EmitParams p = pin;
    const int img = blockIdx.x;
    {
        const size_t o = (size_t)img * p.img_ws;
        p.dets = (const float*)((const char*)p.dets + o);
        p.keep = (const int*)((const char*)p.keep + o);
        p.n_keep = (const int*)((const char*)p.n_keep + o);
        p.dets_out += (size_t)img * p.out_cap * (4 * p.T + 2);
        p.kp_rois += (size_t)img * p.out_cap * (4 * p.T + 1);
        p.n_out += 2 * img;
    }
    const double im_scale = dat_pick(pin.im_scale, img);
    __shared__ unsigned scan[SEL_THREADS];
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_need;
    const int tid = threadIdx.x;
    const int cols = 4 * p.T + 1;
    const int nc = p.K - 1;
    int total = 0;
    for (int c = 0; c < nc; ++c) total += p.n_keep[c];
    unsigned thr_bits = 0u;      // keep everything
    if (p.D > 0 && total > p.D) {
        if (tid == 0) { s_prefix = 0u; s_need = (unsigned)p.D; }
        __syncthreads();
        for (int d = 3; d >= 0; --d) {
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
            const unsigned prefix = s_prefix;
            const int shift = 8 * d;
            const unsigned himask = d == 3 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (int c = 0; c < nc; ++c) {
                const int m = p.n_keep[c];
                for (int i = tid; i < m; i += SEL_THREADS) {
                    const int row = p.keep[(size_t)c * p.cap + i];
                    const unsigned v = __float_as_uint(p.dets[((size_t)c * p.cap + row) * cols + 4 * p.T]);
                    if ((v & himask) == (prefix & himask)) atomicAdd(&hist[(v >> shift) & 255u], 1u);
                }
            }
            __syncthreads();
            if (tid == 0) {
                unsigned rem = s_need, cum = 0;
                int dig = 255;
                for (; dig >= 0; --dig) {
                    if (cum + hist[dig] >= rem) break;
                    cum += hist[dig];
                }
                s_need = rem - cum;
                s_prefix = prefix | ((unsigned)dig << shift);
            }
            __syncthreads();
        }
        thr_bits = s_prefix;     // the D-th largest score
    }
    // ordered emit: classes in order, kept rows in NMS output order
    unsigned base = 0;
    for (int c = 0; c < nc; ++c) {
        const int m = p.n_keep[c];
        const int per = (m + SEL_THREADS - 1) / SEL_THREADS;
        const int lo = tid * per, hi = min(m, lo + per);
        unsigned local = 0;
        for (int i = lo; i < hi; ++i) {
            const int row = p.keep[(size_t)c * p.cap + i];
            local += __float_as_uint(p.dets[((size_t)c * p.cap + row) * cols + 4 * p.T]) >= thr_bits ? 1u : 0u;
        }
        unsigned tot;
        unsigned pos = base + block_scan(local, scan, &tot);
        for (int i = lo; i < hi; ++i) {
            const int row = p.keep[(size_t)c * p.cap + i];
            const float* src = p.dets + ((size_t)c * p.cap + row) * cols;
            if (__float_as_uint(src[4 * p.T]) >= thr_bits) {
                if ((int)pos < p.out_cap) {
                    float* o = p.dets_out + (size_t)pos * (cols + 1);
                    float* k = p.kp_rois + (size_t)pos * cols;
                    k[0] = (float)img;   // single scale: pyramid level 0 (test.py:106-123) of image `img` of the batch
                    for (int q = 0; q < 4 * p.T; ++q) {
                        o[q] = src[q];
                        k[1 + q] = (float)((double)src[q] * im_scale);
                    }
                    o[4 * p.T] = src[4 * p.T];
                    o[4 * p.T + 1] = (float)(c + 1);
                }
                ++pos;
            }
        }
        base += tot;
    }
    // rows past the kept ones: zero rois (the keypoint net runs on all out_cap rows)
    for (int i = (int)base * cols + tid; i < p.out_cap * cols; i += SEL_THREADS) p.kp_rois[i] = 0.f;
    for (int i = (int)base * (cols + 1) + tid; i < p.out_cap * (cols + 1); i += SEL_THREADS) p.dets_out[i] = 0.f;
    if (tid == 0) {
        p.n_out[0] = min((int)base, p.out_cap);
        p.n_out[1] = (int)base;
    }
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

size_t dat_box_results_workspace_bytes(int roi_cap, int num_classes, int T) {
    const int nc = num_classes > 1 ? num_classes - 1 : 1;
    return align_up((size_t)nc * roi_cap * (4 * T + 1) * 4) + align_up((size_t)nc * roi_cap * 4) + 2 * align_up((size_t)nc * 4) +
           dat_nms_ws_bytes(roi_cap, T);
}

int dat_box_results_batch(dat_ctx* ctx, dat_stream s, const float* rois, const int* n_rois, int roi_cap, const float* cls_prob,
                          int prob_ld, const float* bbox_pred, int pred_ld, const dat_det_desc* d, int n_images, void* workspace,
                          int out_cap, float* dets_out, float* keypoint_rois, int* n_out) {
    DAT_ENFORCE(ctx, rois && n_rois && cls_prob && bbox_pred && d && workspace && dets_out && keypoint_rois && n_out,
                "box_results: null argument");
    DAT_ENFORCE(ctx, n_images >= 1 && n_images <= MAX_IMAGES, "box_results: %d images per launch (1..%d)", n_images, MAX_IMAGES);
    DAT_ENFORCE(ctx, d->num_classes >= 2 && d->T >= 1 && d->T <= DET_MAX_T, "box_results: %d classes / T %d unsupported", d->num_classes, d->T);
    DAT_ENFORCE(ctx, roi_cap > 0 && roi_cap <= 16384 && out_cap > 0, "box_results: roi capacity %d / output capacity %d out of range", roi_cap, out_cap);
    DAT_ENFORCE(ctx, prob_ld >= d->num_classes && pred_ld >= (d->cls_agnostic_bbox_reg ? 4 * d->T : d->num_classes * 4 * d->T),
                "box_results: row strides %d / %d too small for %d classes x T %d", prob_ld, pred_ld, d->num_classes, d->T);
    hipStream_t st = (hipStream_t)s;
    const int nc = d->num_classes - 1, T = d->T, cols = 4 * T + 1;
    const size_t img_ws = dat_box_results_workspace_bytes(roi_cap, d->num_classes, T);
    char* ws = (char*)workspace;
    size_t off = 0;
    float* dets = (float*)(ws + off); off += align_up((size_t)nc * roi_cap * cols * 4);
    int* keep = (int*)(ws + off); off += align_up((size_t)nc * roi_cap * 4);
    int* n_sel = (int*)(ws + off); off += align_up((size_t)nc * 4);
    int* n_keep = (int*)(ws + off); off += align_up((size_t)nc * 4);
    char* nms_ws = ws + off;
    DetParams p;
    memset(&p, 0, sizeof(p));
    p.rois = rois; p.n_rois = n_rois; p.prob = cls_prob; p.pred = bbox_pred;
    p.prob_ld = prob_ld; p.pred_ld = pred_ld; p.roi_cap = roi_cap; p.T = T; p.K = d->num_classes;
    p.cls_agnostic = d->cls_agnostic_bbox_reg;
    p.wx = d->reg_weights[0]; p.wy = d->reg_weights[1]; p.ww = d->reg_weights[2]; p.wh = d->reg_weights[3];
    p.xform_clip = d->xform_clip; p.score_thresh = d->score_thresh;
    p.img_ws = img_ws;
    EmitParams e;
    memset(&e, 0, sizeof(e));
    for (int i = 0; i < n_images; ++i) {
        DAT_ENFORCE(ctx, d[i].im_scale > 0.f && d[i].nms_thresh > 0.f, "box_results: im_scale and TEST.NMS must be > 0");
        p.im_scale[i] = e.im_scale[i] = (double)d[i].im_scale_f64;
        p.im_h[i] = (float)d[i].im_h; p.im_w[i] = (float)d[i].im_w;
    }
    const unsigned ni = (unsigned)n_images;
    for (int c = 0; c < nc; ++c) {
        float* dets_c = dets + (size_t)c * roi_cap * cols;
        if (p.T == 1) hipLaunchKernelGGL(det_select_kernel<1>, dim3(ni), dim3(SEL_THREADS), 0, st, p, c + 1, dets_c, n_sel + c);
        else hipLaunchKernelGGL(det_select_kernel<DET_MAX_T>, dim3(ni), dim3(SEL_THREADS), 0, st, p, c + 1, dets_c, n_sel + c);
        int rc = dat_nms_impl_batch(ctx, st, nms_ws, img_ws, dets_c, img_ws / 4, 0, n_sel + c, (int)(img_ws / 4), roi_cap, T, d->nms_thresh,
                                    0, 0, keep + (size_t)c * roi_cap, (int)(img_ws / 4), n_keep + c, (int)(img_ws / 4), n_images);
        if (rc != DAT_OK) return rc;
    }
    e.dets = dets; e.keep = keep; e.n_keep = n_keep; e.K = d->num_classes; e.T = T; e.cap = roi_cap; e.D = d->detections_per_im;
    e.out_cap = out_cap; e.img_ws = img_ws; e.dets_out = dets_out; e.kp_rois = keypoint_rois; e.n_out = n_out;
    hipLaunchKernelGGL(det_limit_emit_kernel, dim3(ni), dim3(SEL_THREADS), 0, st, e);
    DAT_CHECK_LAUNCH(ctx, "box_results");
    return DAT_OK;
}

int dat_box_results(dat_ctx* ctx, dat_stream s, const float* rois, const int* n_rois, int roi_cap, const float* cls_prob, int prob_ld,
                    const float* bbox_pred, int pred_ld, const dat_det_desc* d, void* workspace, int out_cap, float* dets_out,
                    float* keypoint_rois, int* n_out) {
    return dat_box_results_batch(ctx, s, rois, n_rois, roi_cap, cls_prob, prob_ld, bbox_pred, pred_ld, d, 1, workspace, out_cap, dets_out,
                                 keypoint_rois, n_out);
}

// Soft-NMS on the HOST (lib/utils/cython_nms.pyx:98-203; caller lib/core/nms_wrapper.py:29-46, lib/core/test.py:766-772): the
// greedy in-place re-scoring loop in C float arithmetic, statement for statement what Cython emits for the reference's .pyx
// (differences in float, `+ 1` and the area products in double, the Gaussian weight through a double exp, the
// discard-by-swap-with-last that makes the result order-dependent).  Plain host code: the algorithm is sequential by construction and off in every shipped config.
int dat_soft_nms_host(const float* boxes_in, int n, float sigma, float Nt, float threshold, int method, float* boxes_out,
                      int* inds_out, int* n_out) {
    if (!boxes_in || !boxes_out || !inds_out || !n_out || n < 0 || method < 0 || method > 2) return DAT_ERR_ARG;
    float* b = boxes_out;
    memcpy(b, boxes_in, (size_t)n * 5 * sizeof(float));
    for (int i = 0; i < n; ++i) inds_out[i] = i;
    int N = n;
    for (int i = 0; i < N; ++i) {
        float maxscore = b[i * 5 + 4];
        int maxpos = i;
        float tx1 = b[i * 5 + 0], ty1 = b[i * 5 + 1], tx2 = b[i * 5 + 2], ty2 = b[i * 5 + 3], ts = b[i * 5 + 4];
        const int ti = inds_out[i];
        for (int pos = i + 1; pos < N; ++pos)
            if (maxscore < b[pos * 5 + 4]) { maxscore = b[pos * 5 + 4]; maxpos = pos; }
        for (int c = 0; c < 5; ++c) b[i * 5 + c] = b[maxpos * 5 + c];
        inds_out[i] = inds_out[maxpos];
        b[maxpos * 5 + 0] = tx1; b[maxpos * 5 + 1] = ty1; b[maxpos * 5 + 2] = tx2; b[maxpos * 5 + 3] = ty2; b[maxpos * 5 + 4] = ts;
        inds_out[maxpos] = ti;
        tx1 = b[i * 5 + 0]; ty1 = b[i * 5 + 1]; tx2 = b[i * 5 + 2]; ty2 = b[i * 5 + 3];
        int pos = i + 1;
        while (pos < N) {
            const float x1 = b[pos * 5 + 0], y1 = b[pos * 5 + 1], x2 = b[pos * 5 + 2], y2 = b[pos * 5 + 3];
            // (`+ 1` is a DOUBLE constant in the C that Cython emits for the .pyx: sums and the area products are formed in double and
            //  rounded once to float where the .pyx assigns a `cdef float`)
            const float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
            const float iw = (float)((double)(fminf(tx2, x2) - fmaxf(tx1, x1)) + 1.0);
            if (iw > 0) {
                const float ih = (float)((double)(fminf(ty2, y2) - fmaxf(ty1, y1)) + 1.0);
                if (ih > 0) {
                    const float ua = (float)((((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0) + (double)area) - (double)(iw * ih));
                    const float ov = iw * ih / ua;
                    float weight;
                    if (method == 1) weight = ov > Nt ? (float)(1.0 - (double)ov) : 1.f;
                    else if (method == 2) weight = (float)exp((double)(-(ov * ov) / sigma));
                    else weight = ov > Nt ? 0.f : 1.f;
                    b[pos * 5 + 4] = weight * b[pos * 5 + 4];
                    if (b[pos * 5 + 4] < threshold) {   // discard: swap with the last box, shrink N, revisit this slot
                        for (int c = 0; c < 5; ++c) b[pos * 5 + c] = b[(N - 1) * 5 + c];
                        inds_out[pos] = inds_out[N - 1];
                        --N;
                        --pos;
                    }
                }
            }
            ++pos;
        }
    }
    *n_out = N;
    return DAT_OK;
}

}  // extern "C"
