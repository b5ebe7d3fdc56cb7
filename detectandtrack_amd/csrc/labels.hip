// RPN anchor labelling, device half (training input pipeline).
//
// The reference labels every anchor of the field on the host (lib/roi_data/rpn.py:254-370): IoU of up to ~450 k anchors
// against the clip's ground-truth tubes with the Cython kernel (lib/utils/cython_bbox.pyx:16-57, averaged over the
// tube's frames by lib/utils/boxes.py:60-69), a row arg-max, a column max, and "every gt keeps its best anchors".
// That O(anchors x gts) part runs here, one thread per anchor, in the float/double evaluation order of the C that
// Cython emits so that the thresholded labels are identical; the host keeps only the two random sub-samplings.
// This file is compiled with -ffp-contract=off.
#include "dat_common.h"

namespace {

// IoU of one frame's anchor box against one gt box (cython_bbox.pyx:33-56): float differences, `+ 1` in double,
// the union formed in double and rounded once, float product and quotient.
__device__ __forceinline__ float box_iou(const float* __restrict__ b, const float* __restrict__ q) {
    const float iw = (float)((double)(fminf(b[2], q[2]) - fmaxf(b[0], q[0])) + 1.0);
    const float ih = (float)((double)(fminf(b[3], q[3]) - fmaxf(b[1], q[1])) + 1.0);
    if (!(iw > 0.f) || !(ih > 0.f)) return 0.f;
    const double bw = (double)(b[2] - b[0]) + 1.0, bh = (double)(b[3] - b[1]) + 1.0;
    const float qa = (float)(((double)(q[2] - q[0]) + 1.0) * ((double)(q[3] - q[1]) + 1.0));
    const float inter = iw * ih;
    const float ua = (float)(bw * bh + (double)qa - (double)inter);
    return inter / ua;
}

// mean over the tube's frames: float adds in frame order, one float division (np.mean over the per-frame matrices)
template <int MAXT>
__device__ __forceinline__ float tube_iou(const float* __restrict__ a, const float* __restrict__ q, int T) {
    float acc = box_iou(a, q);
    for (int t = 1; t < T; ++t) acc += box_iou(a + 4 * t, q + 4 * t);
    return T == 1 ? acc : acc / (float)T;
}

constexpr int LAB_MAXT = 8;     // frames per tube held in registers
constexpr int LAB_GT_LDS = 64;  // gts staged in LDS per pass

__device__ __forceinline__ bool anchor_inside(const float* a, int T, float lo, float xmax, float ymax) {
    bool ok = true;
    for (int t = 0; t < T; ++t)
        ok = ok && a[4 * t] >= lo && a[4 * t + 1] >= lo && a[4 * t + 2] < xmax && a[4 * t + 3] < ymax;
    return ok;
}

// pass 1: per anchor max / first arg-max over the gts (-1 / 0 for anchors that straddle the image border), and the
// per-gt maximum over the inside anchors (IoU >= 0, so the unsigned order of the float bits is the float order).
__global__ __launch_bounds__(256) void anchor_overlap_kernel(const float* __restrict__ anchors, int n, const float* __restrict__ gts, int G,
                                                             int T, int use_straddle, float lo, float xmax, float ymax,
                                                             float* __restrict__ a2g_max, int* __restrict__ a2g_arg,
                                                             unsigned int* __restrict__ gt_max) {
    __shared__ float sq[LAB_GT_LDS * 4 * LAB_MAXT];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a[4 * LAB_MAXT];
    bool inside = false;
    if (i < n) {
        for (int c = 0; c < 4 * T; ++c) a[c] = anchors[(size_t)i * 4 * T + c];
        inside = !use_straddle || anchor_inside(a, T, lo, xmax, ymax);
    }
    float best = inside && G > 0 ? -1.f : (inside ? 0.f : -1.f);
    int arg = 0;
    for (int g0 = 0; g0 < G; g0 += LAB_GT_LDS) {
        const int gn = min(LAB_GT_LDS, G - g0);
        __syncthreads();
        for (int e = threadIdx.x; e < gn * 4 * T; e += 256) sq[e] = gts[(size_t)g0 * 4 * T + e];
        __syncthreads();
        for (int g = 0; g < gn; ++g) {
            float v = inside ? tube_iou<LAB_MAXT>(a, sq + g * 4 * T, T) : 0.f;
            if (inside && v > best) { best = v; arg = g0 + g; }
            // wave-wide max, one atomic per wave and gt
            float m = v;
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(gt_max + g0 + g, __float_as_uint(m));
        }
    }
    if (i < n) { a2g_max[i] = best; a2g_arg[i] = arg; }
}

// pass 2: flag the anchors that attain some gt's maximum (rpn.py:300-305; a gt no inside anchor touches has maximum
// 0 and — as in the reference — flags every inside anchor with zero overlap)
__global__ __launch_bounds__(256) void anchor_best_kernel(const float* __restrict__ anchors, int n, const float* __restrict__ gts, int G, int T,
                                                          const float* __restrict__ a2g_max, const unsigned int* __restrict__ gt_max,
                                                          unsigned char* __restrict__ flag) {
    __shared__ float sq[LAB_GT_LDS * 4 * LAB_MAXT];
    __shared__ float sm[LAB_GT_LDS];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a[4 * LAB_MAXT];
    bool inside = false;
    if (i < n) {
        inside = a2g_max[i] >= 0.f;
        for (int c = 0; c < 4 * T; ++c) a[c] = anchors[(size_t)i * 4 * T + c];
    }
    bool hit = false;
    for (int g0 = 0; g0 < G; g0 += LAB_GT_LDS) {
        const int gn = min(LAB_GT_LDS, G - g0);
        __syncthreads();
        for (int e = threadIdx.x; e < gn * 4 * T; e += 256) sq[e] = gts[(size_t)g0 * 4 * T + e];
        for (int e = threadIdx.x; e < gn; e += 256) sm[e] = __uint_as_float(gt_max[g0 + e]);
        __syncthreads();
        if (inside)
            for (int g = 0; g < gn; ++g) hit = hit || tube_iou<LAB_MAXT>(a, sq + g * 4 * T, T) == sm[g];
    }
    if (i < n) flag[i] = hit ? 1 : 0;
}

__global__ void scatter_words_kernel(unsigned int* __restrict__ dst, long long dst_words, const int* __restrict__ offsets,
                                     const unsigned int* __restrict__ values, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const long long o = offsets[i];
        if (o >= 0 && o < dst_words) dst[o] = values[i];
    }
}

}  // namespace

extern "C" {

int dat_anchor_overlaps(dat_ctx* ctx, dat_stream s, const float* anchors, int n, const float* gts, int G, int T, float im_h, float im_w,
                        float straddle, float* a2g_max, int* a2g_arg, unsigned char* best_flag, unsigned int* gt_max) {
    DAT_ENFORCE(ctx, anchors && a2g_max && a2g_arg && best_flag && gt_max, "anchor_overlaps: null argument");
    DAT_ENFORCE(ctx, T >= 1 && T <= LAB_MAXT, "anchor_overlaps: T %d must be 1..%d", T, LAB_MAXT);
    DAT_ENFORCE(ctx, G == 0 || gts, "anchor_overlaps: null gts");
    if (n == 0) return DAT_OK;
    hipStream_t st = (hipStream_t)s;
    if (G > 0 && hipMemsetAsync(gt_max, 0, sizeof(unsigned int) * G, st) != hipSuccess) DAT_FAIL(ctx, DAT_ERR_LAUNCH, "anchor_overlaps: memset failed");
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const int use_straddle = straddle >= 0.f;
    hipLaunchKernelGGL(anchor_overlap_kernel, dim3(blocks), dim3(256), 0, st, anchors, n, gts, G, T, use_straddle, -straddle, im_w + straddle,
                       im_h + straddle, a2g_max, a2g_arg, gt_max);
    DAT_CHECK_LAUNCH(ctx, "anchor_overlaps");
    if (G > 0) {
        hipLaunchKernelGGL(anchor_best_kernel, dim3(blocks), dim3(256), 0, st, anchors, n, gts, G, T, a2g_max, gt_max, best_flag);
        DAT_CHECK_LAUNCH(ctx, "anchor_best");
    } else if (hipMemsetAsync(best_flag, 0, (size_t)n, st) != hipSuccess) {
        DAT_FAIL(ctx, DAT_ERR_LAUNCH, "anchor_overlaps: memset failed");
    }
    return DAT_OK;
}

int dat_scatter_words(dat_ctx* ctx, dat_stream s, void* dst, long long dst_words, const int* offsets, const void* values, int n) {
    DAT_ENFORCE(ctx, dst && (n == 0 || (offsets && values)), "scatter_words: null argument");
    if (n == 0) return DAT_OK;
    hipLaunchKernelGGL(scatter_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, (unsigned int*)dst, dst_words, offsets,
                       (const unsigned int*)values, n);
    DAT_CHECK_LAUNCH(ctx, "scatter_words");
    return DAT_OK;
}

}  // extern "C"
