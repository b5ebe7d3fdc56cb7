// Network-input preparation on the device (gfx950): uint8 BGR frames -> the `data` blob.
//
// Replaces the host side of lib/utils/blob.py:40-90 (`prep_im_for_blob`: float32 cast, `im - PIXEL_MEANS`, cv2.resize with
// INTER_LINEAR by fx = fy = scale; `im_list_to_blob`: zero padding to a multiple of FPN.COARSEST_STRIDE, HWC -> NC[T]HW,
// lib/utils/image.py:82-93 batch -> time) for frames that are ALREADY in device memory as uint8 HWC: a clip then crosses PCIe as
// 3 bytes per SOURCE pixel (22 MB for eight 720 x 1280 frames) instead of 12 bytes per padded network pixel (99 MB).
//
// Arithmetic = OpenCV 3.4.1's float32 bilinear path in the order of the host restatement (detectandtrack_amd/utils/image.py,
// checked against oracle/resize.py): source coordinate f = float32((d + 0.5) / scale - 0.5) formed in double, horizontal pass
// first (columns outside the image snap onto the border pixel with weights (1, 0)), then the vertical blend (weights kept, row
// indices clamped); every product and sum rounded to float32 -- this file is compiled with -ffp-contract=off.  The mean is
// subtracted in double and rounded once: `im.astype(float32) - PIXEL_MEANS` is a float64 array in NumPy (PIXEL_MEANS is
// float64), which the resize then casts to float32.  Bit-identical to the host path (tests/test_gpu_kernels.py).
#include "dat_common.h"

namespace {

struct PreParams {
    const uint8_t* frames;   // [F][h][w][3]
    float* data;             // [F / T][3][T][ph][pw]
    int F, T, h, w, oh, ow, ph, pw;
    double inv_fx, inv_fy;   // 1 / scale
    double mean[3];
};

__device__ __forceinline__ void src_coord(int d, double inv_scale, int n, bool snap, int* i0, int* i1, float* t) {
    const float f = (float)(((double)d + 0.5) * inv_scale - 0.5);
    const float fl = floorf(f);
    int s = (int)fl;
    float tt = f - fl;
    if (snap) {                     // x set-up loop of cv::resize: out-of-range columns read the border pixel with weights (1, 0)
        if (s < 0 || s >= n - 1) tt = 0.f;
        s = min(max(s, 0), n - 1);
        *i0 = s;
        *i1 = min(s + 1, n - 1);
    } else {                        // y: weights kept, the two row indices clamped
        *i0 = min(max(s, 0), n - 1);
        *i1 = min(max(s + 1, 0), n - 1);
    }
    *t = tt;
}

// One thread = PX horizontally adjacent output pixels (round 4): the one-pixel-per-thread form launched 147 k blocks whose threads
// each waited out one round of twelve single-byte loads before three 4-byte stores (0.54 ms for a 4-clip batch, 0.9 TB/s of its
// 484 MB); with four pixels the 48 loads of a thread are in flight together and every plane receives one 16-byte store.  The
// per-pixel arithmetic is untouched (bit-identical output).
constexpr int PX = 8;
__global__ __launch_bounds__(256) void preprocess_kernel(const PreParams p) {
    const int xb = (blockIdx.x * 256 + threadIdx.x) * PX;
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    if (xb >= p.pw) return;
    const int n = f / p.T, t = f - n * p.T;
    const size_t plane = (size_t)p.ph * p.pw;
    float* out = p.data + (((size_t)n * 3) * p.T + t) * plane + (size_t)y * p.pw + xb;
    const size_t cstep = (size_t)p.T * plane;
    float v[3][PX];
    const bool row_live = y < p.oh;
    int y0 = 0, y1 = 0;
    float ty = 0.f;
    if (row_live) src_coord(y, p.inv_fy, p.h, false, &y0, &y1, &ty);
    const uint8_t* fr = p.frames + (size_t)f * p.h * p.w * 3;
    const uint8_t* r0 = fr + (size_t)y0 * p.w * 3;
    const uint8_t* r1 = fr + (size_t)y1 * p.w * 3;
    const float uy = 1.f - ty;
    // ---- all loads first (clamped addresses: unconditional, independent) ----
    int x0[PX], x1[PX];
    float tx[PX];
    uint8_t q[PX][4][3];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int x = min(xb + i, p.ow - 1);          // (columns in the padding are computed on a clamped x and overwritten with 0 below)
        src_coord(x, p.inv_fx, p.w, true, &x0[i], &x1[i], &tx[i]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            q[i][0][c] = r0[x0[i] * 3 + c]; q[i][1][c] = r0[x1[i] * 3 + c];
            q[i][2][c] = r1[x0[i] * 3 + c]; q[i][3][c] = r1[x1[i] * 3 + c];
        }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const float ux = 1.f - tx[i];
        const bool live = row_live && xb + i < p.ow;   // else: zero padding up to the stride multiple (blob.py:47-55)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a00 = (float)((double)q[i][0][c] - p.mean[c]), a01 = (float)((double)q[i][1][c] - p.mean[c]);
            const float a10 = (float)((double)q[i][2][c] - p.mean[c]), a11 = (float)((double)q[i][3][c] - p.mean[c]);
            const float top = a00 * ux + a01 * tx[i];      // horizontal pass (two rounded products, one rounded sum)
            const float bot = a10 * ux + a11 * tx[i];
            v[c][i] = live ? top * uy + bot * ty : 0.f;    // vertical blend
        }
    }
    if (xb + PX <= p.pw && (p.pw & 3) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < PX; i += 4) *(float4*)(out + c * cstep + i) = make_float4(v[c][i], v[c][i + 1], v[c][i + 2], v[c][i + 3]);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            for (int i = 0; i < PX && xb + i < p.pw; ++i) out[c * cstep + i] = v[c][i];
    }
}

}  // namespace

extern "C" {

int dat_preprocess_frames(dat_ctx* ctx, dat_stream s, const unsigned char* frames, int n_frames, int T, int h, int w, double fx, double fy,
                          int out_h, int out_w, int pad_h, int pad_w, const double* pixel_means, float* data) {
    DAT_ENFORCE(ctx, frames && data && pixel_means, "preprocess_frames: null argument");
    DAT_ENFORCE(ctx, n_frames > 0 && T > 0 && n_frames % T == 0, "preprocess_frames: %d frames are not whole clips of %d", n_frames, T);
    DAT_ENFORCE(ctx, h > 0 && w > 0 && out_h > 0 && out_w > 0 && pad_h >= out_h && pad_w >= out_w && fx > 0 && fy > 0,
                "preprocess_frames: bad geometry %dx%d -> %dx%d (pad %dx%d)", h, w, out_h, out_w, pad_h, pad_w);
    DAT_ENFORCE(ctx, n_frames <= 65535 && pad_h <= 65535, "preprocess_frames: grid too large");
    PreParams p;
    p.frames = frames; p.data = data; p.F = n_frames; p.T = T; p.h = h; p.w = w; p.oh = out_h; p.ow = out_w; p.ph = pad_h; p.pw = pad_w;
    p.inv_fx = 1.0 / fx; p.inv_fy = 1.0 / fy;
    for (int c = 0; c < 3; ++c) p.mean[c] = pixel_means[c];
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((pad_w + 256 * PX - 1) / (256 * PX)), (unsigned)pad_h, (unsigned)n_frames), dim3(256), 0,
                       (hipStream_t)s, p);
    DAT_CHECK_LAUNCH(ctx, "preprocess_frames");
    return DAT_OK;
}

}  // extern "C"
