// RoIAlign over NDHWC feature maps, tube- and FPN-level aware (gfx950).
//
// One launch replaces the reference's per-head chain (lib/modeling/detector.py:216-310):
//   RoIToBatchFormat (host numpy, ops/roi_blob_transforms.py:25-36) -> MoveTimeToBatchDim (Transpose+Reshape)
//   -> RoIAlign per FPN level -> Concat -> BatchPermutation (rois_idx_restore) -> MoveTimeToBatchDimInverse.
// Here every RoI picks its FPN level in-kernel (lib/modeling/FPN.py:349-360), reads the frame n*T+t of that
// level directly, and writes its output row in RoI order, so no transposes, concat or permutation exist.
//
// Sampling semantics = legacy Detectron RoIAlign (Caffe2 modules/detectron, restated in oracle/roi_align.py).
// Work split: a group of lanes per (roi, frame, ph, pw) output cell, the lanes striding over the channels in 16-byte
// vectors, so each of the (sampling^2 x 4) bilinear taps is a contiguous channel read — the wave-level bilinear reduction the C
// axis of NDHWC makes natural.  The group is the power of two covering C / (16 bytes) lanes, at most 64: 256 bf16 channels are 32
// vectors, so a wave works on TWO cells at a time and all 64 lanes fetch (one cell per wave left half of them idle).
#include "dat_common.h"

namespace {

struct RoiLevels {
    const char* feat[4];
    int H[4], W[4];
    float scale[4];
    int n_levels, k_min;
    float canon_scale;
    int canon_level;
};

struct RoiParams {
    RoiLevels lv;
    int T, C;
    const float* rois;
    int R, Tr, t0, pooled, sampling;
    char* out;
};

template <int DT>
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiParams p) {
    constexpr int V = 16 / ElemOf<DT>::size;  // channels per 16-byte vector
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int P = p.pooled;
    const long long ncell = (long long)p.R * p.Tr * P * P;
    const int cvecs = p.C / V;
    const int roi_cols = 4 * p.Tr + 1;
    int grp = 64;                                   // lanes per cell
    while (grp > 1 && (grp >> 1) >= cvecs) grp >>= 1;
    const int cpw = 64 / grp;                       // cells per wave
    const int sub = lane / grp, gl = lane - sub * grp;
    for (long long cell = ((long long)blockIdx.x * 4 + wave) * cpw + sub; cell < ncell; cell += (long long)gridDim.x * 4 * cpw) {
        const int pw = cell % P;
        long long q = cell / P;
        const int ph = q % P; q /= P;
        const int t = q % p.Tr;
        const int r = q / p.Tr;
        const float* roi = p.rois + (size_t)r * roi_cols;
        const int n = (int)roi[0];
        const float bx1 = roi[1 + 4 * t], by1 = roi[2 + 4 * t], bx2 = roi[3 + 4 * t], by2 = roi[4 + 4 * t];
        // FPN level (FPN.py:349-360, boxes.py:72-78): mean area over the tube's frames
        int lvl = 0;
        if (p.lv.n_levels > 1) {
            float asum = 0.f;
            for (int tt = 0; tt < p.Tr; ++tt) {
                const float w = roi[3 + 4 * tt] - roi[1 + 4 * tt] + 1.f;
                const float h = roi[4 + 4 * tt] - roi[2 + 4 * tt] + 1.f;
                asum += w * h;
            }
            const float sarea = sqrtf(asum / (float)p.Tr);
            float l = floorf((float)p.lv.canon_level + log2f(sarea / p.lv.canon_scale + 1e-6f));
            const int k_max = p.lv.k_min + p.lv.n_levels - 1;
            l = fminf(fmaxf(l, (float)p.lv.k_min), (float)k_max);
            lvl = (int)l - p.lv.k_min;
        }
        const int H = p.lv.H[lvl], W = p.lv.W[lvl];
        const float sc = p.lv.scale[lvl];
        const int frame = n * p.T + (p.Tr == 1 ? p.t0 : t);
        const char* fbase = p.lv.feat[lvl] + (size_t)frame * H * W * p.C * ElemOf<DT>::size;

        const float x1 = bx1 * sc, y1 = by1 * sc, x2 = bx2 * sc, y2 = by2 * sc;
        const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
        const float bw = rw / (float)P, bh = rh / (float)P;
        const int gh = p.sampling > 0 ? p.sampling : (int)ceilf(rh / (float)P);
        const int gw = p.sampling > 0 ? p.sampling : (int)ceilf(rw / (float)P);
        const float inv = 1.f / (float)(gh * gw);

        for (int cv = gl; cv < cvecs; cv += grp) {
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
            for (int iy = 0; iy < gh; ++iy) {
                float y = y1 + (float)ph * bh + ((float)iy + .5f) * bh / (float)gh;
                for (int ix = 0; ix < gw; ++ix) {
                    float x = x1 + (float)pw * bw + ((float)ix + .5f) * bw / (float)gw;
                    float yy = y;
                    if (yy < -1.f || yy > (float)H || x < -1.f || x > (float)W) continue;
                    if (yy <= 0.f) yy = 0.f;
                    if (x <= 0.f) x = 0.f;
                    int yl = (int)yy, xl = (int)x, yh, xh;
                    if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
                    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                    const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                    const size_t co = (size_t)cv * 16;
                    const uint4 v1 = *(const uint4*)(fbase + ((size_t)yl * W + xl) * p.C * ElemOf<DT>::size + co);
                    const uint4 v2 = *(const uint4*)(fbase + ((size_t)yl * W + xh) * p.C * ElemOf<DT>::size + co);
                    const uint4 v3 = *(const uint4*)(fbase + ((size_t)yh * W + xl) * p.C * ElemOf<DT>::size + co);
                    const uint4 v4 = *(const uint4*)(fbase + ((size_t)yh * W + xh) * p.C * ElemOf<DT>::size + co);
                    const uint32_t a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w};
                    const uint32_t a3[4] = {v3.x, v3.y, v3.z, v3.w}, a4[4] = {v4.x, v4.y, v4.z, v4.w};
                    if (DT == DAT_BF16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[(2 * e) % V] += w1 * bf2f((uint16_t)(a1[e] & 0xffff)) + w2 * bf2f((uint16_t)(a2[e] & 0xffff)) +
                                                w3 * bf2f((uint16_t)(a3[e] & 0xffff)) + w4 * bf2f((uint16_t)(a4[e] & 0xffff));
                            acc[(2 * e + 1) % V] += w1 * bf2f((uint16_t)(a1[e] >> 16)) + w2 * bf2f((uint16_t)(a2[e] >> 16)) +
                                                    w3 * bf2f((uint16_t)(a3[e] >> 16)) + w4 * bf2f((uint16_t)(a4[e] >> 16));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[e] += w1 * __uint_as_float(a1[e]) + w2 * __uint_as_float(a2[e]) + w3 * __uint_as_float(a3[e]) +
                                      w4 * __uint_as_float(a4[e]);
                    }
                }
            }
            uint4 o;
            if (DT == DAT_BF16) {
                o.x = f2bf2(acc[0] * inv, acc[1] * inv);
                o.y = f2bf2(acc[2] * inv, acc[3] * inv);
                o.z = f2bf2(acc[4 % V] * inv, acc[5 % V] * inv);
                o.w = f2bf2(acc[6 % V] * inv, acc[7 % V] * inv);
            } else {
                o.x = __float_as_uint(acc[0] * inv); o.y = __float_as_uint(acc[1] * inv);
                o.z = __float_as_uint(acc[2] * inv); o.w = __float_as_uint(acc[3] * inv);
            }
            // out [R*Tr, P, P, C]
            *(uint4*)(p.out + ((((size_t)r * p.Tr + t) * P + ph) * P + pw) * p.C * ElemOf<DT>::size + (size_t)cv * 16) = o;
        }
    }
}

}  // namespace

extern "C" int dat_roi_align(dat_ctx* ctx, dat_stream s, int dtype, const dat_roi_level* levels, int n_levels, int k_min,
                             float canon_scale, int canon_level, int T, int C, const float* rois, int R, int Tr, int t0,
                             int pooled, int sampling_ratio, void* out) {
    DAT_ENFORCE(ctx, levels && n_levels >= 1 && n_levels <= 4, "roi_align: n_levels %d must be 1..4", n_levels);
    DAT_ENFORCE(ctx, C % 8 == 0, "roi_align: C=%d must be a multiple of 8", C);
    DAT_ENFORCE(ctx, Tr == 1 || Tr == T, "roi_align: tube length %d must be 1 or the feature T %d", Tr, T);
    if (R == 0) return DAT_OK;
    DAT_ENFORCE(ctx, rois && out, "roi_align: null argument");
    RoiParams p;
    memset(&p, 0, sizeof(p));
    for (int i = 0; i < n_levels; ++i) {
        p.lv.feat[i] = (const char*)levels[i].feat;
        p.lv.H[i] = levels[i].H;
        p.lv.W[i] = levels[i].W;
        p.lv.scale[i] = levels[i].spatial_scale;
    }
    p.lv.n_levels = n_levels; p.lv.k_min = k_min; p.lv.canon_scale = canon_scale; p.lv.canon_level = canon_level;
    p.T = T; p.C = C; p.rois = rois; p.R = R; p.Tr = Tr; p.t0 = t0; p.pooled = pooled; p.sampling = sampling_ratio;
    p.out = (char*)out;
    const long long ncell = (long long)R * Tr * pooled * pooled;
    const int cvecs = C / (16 / dat_esize(dtype));
    int grp = 64;
    while (grp > 1 && (grp >> 1) >= cvecs) grp >>= 1;
    const int cells_per_block = 4 * (64 / grp);
    long long blocks = (ncell + cells_per_block - 1) / cells_per_block;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (dtype == DAT_BF16)
        hipLaunchKernelGGL(roi_align_kernel<DAT_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p);
    else
        hipLaunchKernelGGL(roi_align_kernel<DAT_F32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p);
    DAT_CHECK_LAUNCH(ctx, "roi_align");
    return DAT_OK;
}
