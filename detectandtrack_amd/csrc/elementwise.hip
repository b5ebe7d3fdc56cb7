// HBM-bound elementwise / layout kernels of the hot path (gfx950).
// Each is one pass: read-once + write-once of the logical tensor; 16-byte accesses where the layout allows.
#include <algorithm>
#include <stdlib.h>

#include "dat_common.h"

namespace {

constexpr int TPB = 256;
inline int grid_for(size_t n, int per_thread = 1) {
    size_t b = (n + (size_t)TPB * per_thread - 1) / ((size_t)TPB * per_thread);
    const size_t cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride beyond
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---- ZeroEven (lib/ops/zero_even_op.cu:25-30): data[2i] = 0 --------------------------------------
__global__ void zero_even_kernel(float* x, long long n) {
    const long long half = (n + 1) / 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < half; i += (long long)gridDim.x * blockDim.x)
        x[2 * i] = 0.f;
}

// ---- AffineChannelNd (lib/ops/affine_channel_nd_op.cu:20-46) ---------------------------------------
// out[i] = in[i]*scale[(i/inner)%C] + bias[...]; float4 path when inner % 4 == 0 (a float4 never straddles channels)
template <bool HAS_BIAS>
__global__ void affine_nd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                 const float* __restrict__ bias, float* __restrict__ y, int C, long long inner,
                                 long long total) {
    if ((inner & 3) == 0) {
        const long long n4 = total >> 2, inner4 = inner >> 2;
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
            const int c = (int)((i / inner4) % C);
            const float s = scale[c];
            const float b = HAS_BIAS ? bias[c] : 0.f;
            float4 v = ((const float4*)x)[i];
            v.x = v.x * s + b; v.y = v.y * s + b; v.z = v.z * s + b; v.w = v.w * s + b;
            ((float4*)y)[i] = v;
        }
    } else {
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const int c = (int)((i / inner) % C);
            y[i] = x[i] * scale[c] + (HAS_BIAS ? bias[c] : 0.f);
        }
    }
}

// ---- NC(T)HW fp32 <-> [N*T,H,W,Cs] ----------------------------------------------------------------------
// 32x32 LDS transpose tiles over (C, H*W) so that both the read (along HW) and the write (along C) coalesce.
template <int DT>
__global__ void ncdhw_to_ndhwc_kernel(const float* __restrict__ src, void* __restrict__ dst, int N, int C, int T,
                                      int HW, int Cs) {
    __shared__ float tile[32][33];
    const int frame = blockIdx.z;  // n*T + t
    const int n = frame / T, t = frame % T;
    const int hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, hw = hw0 + tx;
        float v = 0.f;
        if (c < C && hw < HW) v = src[(((size_t)n * C + c) * T + t) * HW + hw];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int hw = hw0 + r, c = c0 + tx;
        if (hw < HW && c < Cs) ElemOf<DT>::st(dst, ((size_t)frame * HW + hw) * Cs + c, tile[tx][r]);
    }
}

template <int DT>
__global__ void ndhwc_to_ncdhw_kernel(const void* __restrict__ src, float* __restrict__ dst, int N, int C, int T, int HW,
                                      int Cs) {
    __shared__ float tile[32][33];
    const int frame = blockIdx.z;
    const int n = frame / T, t = frame % T;
    const int hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int hw = hw0 + r, c = c0 + tx;
        float v = 0.f;
        if (hw < HW && c < C) v = ElemOf<DT>::ld(src, ((size_t)frame * HW + hw) * Cs + c);
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, hw = hw0 + tx;
        if (c < C && hw < HW) dst[(((size_t)n * C + c) * T + t) * HW + hw] = tile[tx][r];
    }
}

// ---- MaxPool [1,k,k] on NDHWC; pad cells behave as -inf (Caffe2 MaxPool) ---------------------------------------
// one thread = 8 (bf16) or 4 (fp32) channels of one output position: 16-byte loads/stores
template <int DT>
__global__ void maxpool_hw_kernel(const void* __restrict__ x, void* __restrict__ y, int frames, int H, int W, int C, int Ho,
                                  int Wo, int k, int stride, int pad) {
    constexpr int V = 16 / ElemOf<DT>::size;
    const int cv = C / V;
    const size_t total = (size_t)frames * Ho * Wo * cv;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = i % cv;
        size_t q = i / cv;
        const int ow = q % Wo; q /= Wo;
        const int oh = q % Ho;
        const int f = q / Ho;
        float m[V];
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] = -INFINITY;
        for (int kh = 0; kh < k; ++kh) {
            const int ih = oh * stride - pad + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int iw = ow * stride - pad + kw;
                if (iw < 0 || iw >= W) continue;
                const uint4 v = *(const uint4*)((const char*)x + ((((size_t)f * H + ih) * W + iw) * C + (size_t)cg * V) * ElemOf<DT>::size);
                if (DT == DAT_BF16) {
                    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        m[2 * e] = fmaxf(m[2 * e], bf2f((uint16_t)(u[e] & 0xffff)));
                        m[2 * e + 1] = fmaxf(m[2 * e + 1], bf2f((uint16_t)(u[e] >> 16)));
                    }
                } else {
                    m[0] = fmaxf(m[0], __uint_as_float(v.x)); m[1] = fmaxf(m[1], __uint_as_float(v.y));
                    m[2] = fmaxf(m[2], __uint_as_float(v.z)); m[3] = fmaxf(m[3], __uint_as_float(v.w));
                }
            }
        }
        uint4 o;
        if (DT == DAT_BF16) {
            o.x = f2bf2(m[0], m[1]); o.y = f2bf2(m[2], m[3]);
            o.z = f2bf2(m[4 % V], m[5 % V]); o.w = f2bf2(m[6 % V], m[7 % V]);
        } else {
            o.x = __float_as_uint(m[0]); o.y = __float_as_uint(m[1]); o.z = __float_as_uint(m[2]); o.w = __float_as_uint(m[3]);
        }
        *(uint4*)((char*)y + i * 16) = o;
    }
}

// ---- mean over T: x [N,T,hwc] -> y [N,hwc] (detector.py:559-569 TimePool 'avg') -----------------------------------
template <int DT>
__global__ void time_avg_kernel(const void* __restrict__ x, void* __restrict__ y, int N, int T, long long hwc) {
    const long long total = (long long)N * hwc;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / hwc, r = i - n * hwc;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += ElemOf<DT>::ld(x, ((size_t)n * T + t) * hwc + r);
        ElemOf<DT>::st(y, i, s / (float)T);
    }
}

// ---- mean over H,W: x [frames,HW,Cs] -> y fp32 [frames,C] -----------------------------------------------------------
template <int DT>
__global__ void spatial_mean_kernel(const void* __restrict__ x, float* __restrict__ y, int frames, int HW, int C, int Cs) {
    const int total = frames * C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int f = i / C, c = i - f * C;
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += ElemOf<DT>::ld(x, ((size_t)f * HW + p) * Cs + c);
        y[i] = s / (float)HW;
    }
}

// ---- row softmax over K (tiny K: number of classes) ---------------------------------------------------------------------
__global__ void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int K, int ld_in, int ld_out) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
        const float* xi = x + (size_t)r * ld_in;
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) m = fmaxf(m, xi[k]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(xi[k] - m);
        for (int k = 0; k < K; ++k) y[(size_t)r * ld_out + k] = expf(xi[k] - m) / s;
    }
}

// ---- ConvTranspose k4 s2 p1 weights -> 3x3 sub-pixel conv weights ------------------------------------------------------------
// ConvTranspose: out[Y][X] += in[y][x] * w[ci][k][Y+1-2y][X+1-2x].  For sub-pixel (a,b): Y = 2y'+a, input y = y'+dy
// (dy in -1..1) uses kernel row a+1-2dy when that is in 0..3.  As a cross-correlation 3x3 (pad 1) the tap index is dy+1.
__global__ void deconv_k4s2_weights_kernel(const float* __restrict__ w, int Cin, int K, float* __restrict__ out) {
    const int total = 4 * K * Cin * 9;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int tap = i % 9;
        const int ci = (i / 9) % Cin;
        const int co = i / (9 * Cin);
        const int k = co % K, ab = co / K;
        const int a = ab >> 1, b = ab & 1;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int ky = a + 1 - 2 * dy, kx = b + 1 - 2 * dx;
        float v = 0.f;
        if (ky >= 0 && ky < 4 && kx >= 0 && kx < 4) v = w[(((size_t)ci * K + k) * 4 + ky) * 4 + kx];
        out[i] = v;
    }
}

// ---- pixel shuffle (2x) + fixed bilinear ConvTranspose (k = 2*up, s = up, p = up/2), NCHW fp32 out -------------------------------
// low[Y][X] (Y,X in 0..2S) = sub[Y>>1][X>>1][((Y&1)*2 + (X&1))*K + k]
// out[OY][OX] = sum_{y,x} low[y][x] * f[OY + p - up*y] * f[OX + p - up*x],  f = detector.py:356-366 1-D factor
template <int DT>
__global__ void kps_finalize_kernel(const void* __restrict__ sub, int R, int Tr, int S, int cs, int K, int up,
                                    float* __restrict__ out) {
    const int L = 2 * S, M = L * up;
    const int ksz = 2 * up, pad = up / 2;
    const float factor = (float)((ksz + 1) / 2);
    const float center = (ksz % 2 == 1) ? factor - 1.f : factor - 0.5f;
    const size_t total = (size_t)R * Tr * K * M * M;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = i % M;
        size_t q = i / M;
        const int oy = q % M; q /= M;
        const int k = q % K; q /= K;
        const int t = q % Tr;
        const int r = q / Tr;
        // contributing low-res rows: ky = oy + pad - up*y in [0, ksz)
        float acc = 0.f;
        const int y_hi = (oy + pad) / up;
        const int x_hi = (ox + pad) / up;
        for (int y = y_hi; y >= 0 && oy + pad - up * y < ksz; --y) {
            if (y >= L) continue;
            const int ky = oy + pad - up * y;
            const float fy = 1.f - fabsf((float)ky - center) / factor;
            for (int x = x_hi; x >= 0 && ox + pad - up * x < ksz; --x) {
                if (x >= L) continue;
                const int kx = ox + pad - up * x;
                const float fx = 1.f - fabsf((float)kx - center) / factor;
                const size_t fr = (size_t)r * Tr + t;
                const float v = ElemOf<DT>::ld(sub, ((fr * S + (y >> 1)) * S + (x >> 1)) * cs + ((y & 1) * 2 + (x & 1)) * K + k);
                acc += v * (fy * fx);
            }
        }
        out[i] = acc;  // [R, Tr*K, M, M] with channel t*K + k
    }
}

// The same op with one block per output MAP (roi, frame, keypoint): kps_finalize_kernel spends its time on index arithmetic -- six
// 64-bit divisions and a strided gather per OUTPUT element (21 M of them per 4-clip forward: 200 us for 85 MB written, ~5x its HBM
// time).  Here a block unfolds the map's four sub-pixel channels into LDS as the low-resolution map [2S][2S] (fp32); thread
// (ox, row group) then walks its output column with the x taps fixed, rows in a loop without a division, and the block's ng x M threads
// store ng complete output rows per step.  Same taps, same order, same fp32 expressions: bit-identical to kps_finalize_kernel.
template <int DT>
__global__ void kps_finalize_tile_kernel(const void* __restrict__ sub, int S, int cs, int K, int up, float* __restrict__ out) {
    extern __shared__ float low[];                 // [L][L]
    const int L = 2 * S, M = L * up;
    const int ksz = 2 * up, pad = up / 2;
    const float factor = (float)((ksz + 1) / 2);
    const float center = (ksz % 2 == 1) ? factor - 1.f : factor - 0.5f;
    const int fr = blockIdx.x / K, k = blockIdx.x - fr * K;      // fr = r * Tr + t; output channel t * K + k of roi r = map fr * K + k
    for (int i = threadIdx.x; i < S * S * 4; i += blockDim.x) {
        const int cell = i >> 2, ab = i & 3;
        const int ys = cell / S, xs = cell - ys * S;
        low[(2 * ys + (ab >> 1)) * L + 2 * xs + (ab & 1)] = ElemOf<DT>::ld(sub, ((size_t)fr * S * S + cell) * cs + ab * K + k);
    }
    __syncthreads();
    const int ox = threadIdx.x % M, og = threadIdx.x / M, ng = blockDim.x / M;     // blockDim = ng * M
    const int x_hi = (ox + pad) / up;
    float* const obase = out + (size_t)blockIdx.x * M * M;
    if (up == 2) {
        // the configuration every model uses (KRCNN.UP_SCALE 2): exactly two taps per axis -- x_hi, x_hi - 1 -- visited in the general
        // loop's order; the tap weights of the column are computed once per thread
        const int xa = x_hi, xb = x_hi - 1;
        const bool vxa = xa < L, vxb = xb >= 0;
        const float fxa = 1.f - fabsf((float)(ox + pad - up * xa) - center) / factor;
        const float fxb = 1.f - fabsf((float)(ox + pad - up * xb) - center) / factor;
        for (int oy = og; oy < M; oy += ng) {
            const int ya = (oy + pad) / up, yb = ya - 1;
            const float fya = 1.f - fabsf((float)(oy + pad - up * ya) - center) / factor;
            const float fyb = 1.f - fabsf((float)(oy + pad - up * yb) - center) / factor;
            float acc = 0.f;
            if (ya < L) {
                if (vxa) acc += low[ya * L + xa] * (fya * fxa);
                if (vxb) acc += low[ya * L + xb] * (fya * fxb);
            }
            if (yb >= 0) {
                if (vxa) acc += low[yb * L + xa] * (fyb * fxa);
                if (vxb) acc += low[yb * L + xb] * (fyb * fxb);
            }
            obase[(size_t)oy * M + ox] = acc;
        }
        return;
    }
    for (int oy = og; oy < M; oy += ng) {
        const int y_hi = (oy + pad) / up;
        float acc = 0.f;
        for (int y = y_hi; y >= 0 && oy + pad - up * y < ksz; --y) {
            if (y >= L) continue;
            const int ky = oy + pad - up * y;
            const float fy = 1.f - fabsf((float)ky - center) / factor;
            for (int x = x_hi; x >= 0 && ox + pad - up * x < ksz; --x) {
                if (x >= L) continue;
                const int kx = ox + pad - up * x;
                const float fx = 1.f - fabsf((float)kx - center) / factor;
                acc += low[y * L + x] * (fy * fx);
            }
        }
        obase[(size_t)oy * M + ox] = acc;
    }
}

}  // namespace

#define DISPATCH_DT(dtype, KERNEL, grid, block, st, ...)                                                  \
    do {                                                                                                  \
        if ((dtype) == DAT_BF16)                                                                          \
            hipLaunchKernelGGL((KERNEL<DAT_BF16>), grid, block, 0, st, __VA_ARGS__);                      \
        else                                                                                              \
            hipLaunchKernelGGL((KERNEL<DAT_F32>), grid, block, 0, st, __VA_ARGS__);                       \
    } while (0)

// fp32 NDHWC [npos, C] -> hi / lo bf16 split [npos, 2C]: per 64-channel chunk q the pixel's 128-byte line 2q holds bf16(x), line
// 2q + 1 holds bf16(x - float(bf16(x))) -- x = hi + lo to 2^-17 relative (the difference is exact in fp32).  The operand format of
// the bf16x3 conv mode (conv3d_igemm.hip).  One thread = 8 channels: two 16-byte loads, two 16-byte stores, all fully coalesced.
__global__ void split_bf16x2_kernel(const float* __restrict__ x, uint4* __restrict__ y, long long n8, int c8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const long long pos = i / c8;
        const int g = (int)(i - pos * c8);         // 8-channel group inside the pixel
        const float4 a = *(const float4*)(x + i * 8), b = *(const float4*)(x + i * 8 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = f2bf2(v[2 * e], v[2 * e + 1]);
            const float r0 = v[2 * e] - __uint_as_float(hi[e] << 16), r1 = v[2 * e + 1] - __uint_as_float(hi[e] & 0xffff0000u);
            lo[e] = f2bf2(r0, r1);
        }
        // output in 16-byte units: pixel pos has 2 * c8 of them; chunk q = g / 8 -> hi at unit 16 q + (g & 7), lo 8 units behind
        uint4* dst = y + pos * (2 * c8) + (g >> 3) * 16 + (g & 7);
        dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        dst[8] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

extern "C" {

int dat_zero_even_fwd(dat_ctx* ctx, dat_stream s, float* x, long long n) {
    DAT_ENFORCE(ctx, n >= 0, "ZeroEven: Input(0).ndim() == 1 and size >= 0 required, got n=%lld", n);
    if (n == 0) return DAT_OK;
    DAT_ENFORCE(ctx, x, "ZeroEven: null data");
    hipLaunchKernelGGL(zero_even_kernel, dim3(grid_for((n + 1) / 2)), dim3(TPB), 0, (hipStream_t)s, x, n);
    DAT_CHECK_LAUNCH(ctx, "zero_even");
    return DAT_OK;
}

int dat_affine_channel_nd_fwd(dat_ctx* ctx, dat_stream s, const float* x, const float* scale, const float* bias, float* y,
                              int N, int C, long long inner) {
    DAT_ENFORCE(ctx, x && scale && bias && y, "AffineChannelNd: null argument");
    DAT_ENFORCE(ctx, N >= 0 && C > 0 && inner >= 0, "AffineChannelNd: bad dims N=%d C=%d inner=%lld", N, C, inner);
    const long long total = (long long)N * C * inner;
    if (total == 0) return DAT_OK;
    hipLaunchKernelGGL(affine_nd_kernel<true>, dim3(grid_for(total, 4)), dim3(TPB), 0, (hipStream_t)s, x, scale, bias, y, C,
                       inner, total);
    DAT_CHECK_LAUNCH(ctx, "affine_channel_nd_fwd");
    return DAT_OK;
}

int dat_affine_channel_nd_bwd(dat_ctx* ctx, dat_stream s, const float* dy, const float* scale, float* dx, int N, int C,
                              long long inner) {
    DAT_ENFORCE(ctx, dy && scale && dx, "AffineChannelNdGradient: null argument");
    const long long total = (long long)N * C * inner;
    if (total == 0) return DAT_OK;
    hipLaunchKernelGGL(affine_nd_kernel<false>, dim3(grid_for(total, 4)), dim3(TPB), 0, (hipStream_t)s, dy, scale,
                       (const float*)nullptr, dx, C, inner, total);
    DAT_CHECK_LAUNCH(ctx, "affine_channel_nd_bwd");
    return DAT_OK;
}

int dat_ncdhw_to_ndhwc(dat_ctx* ctx, dat_stream s, const float* src, void* dst, int dtype, int N, int C, int T, int H, int W,
                       int Cs) {
    DAT_ENFORCE(ctx, src && dst && Cs >= C, "ncdhw_to_ndhwc: bad argument");
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (Cs + 31) / 32, N * T);
    DISPATCH_DT(dtype, ncdhw_to_ndhwc_kernel, grid, dim3(256), (hipStream_t)s, src, dst, N, C, T, HW, Cs);
    DAT_CHECK_LAUNCH(ctx, "ncdhw_to_ndhwc");
    return DAT_OK;
}

int dat_ndhwc_to_ncdhw(dat_ctx* ctx, dat_stream s, const void* src, int dtype, float* dst, int N, int C, int T, int H, int W,
                       int Cs) {
    DAT_ENFORCE(ctx, src && dst && Cs >= C, "ndhwc_to_ncdhw: bad argument");
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, N * T);
    DISPATCH_DT(dtype, ndhwc_to_ncdhw_kernel, grid, dim3(256), (hipStream_t)s, src, dst, N, C, T, HW, Cs);
    DAT_CHECK_LAUNCH(ctx, "ndhwc_to_ncdhw");
    return DAT_OK;
}

int dat_maxpool_hw(dat_ctx* ctx, dat_stream s, int dtype, const void* x, void* y, int frames, int H, int W, int C, int k,
                   int stride, int pad) {
    DAT_ENFORCE(ctx, x && y && C % 8 == 0, "maxpool: C=%d must be a multiple of 8", C);
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const size_t total = (size_t)frames * Ho * Wo * (C / (16 / dat_esize(dtype)));
    DISPATCH_DT(dtype, maxpool_hw_kernel, dim3(grid_for(total)), dim3(TPB), (hipStream_t)s, x, y, frames, H, W, C, Ho, Wo, k,
                stride, pad);
    DAT_CHECK_LAUNCH(ctx, "maxpool_hw");
    return DAT_OK;
}

// ---- frame gather / scatter: dst frame dst_idx[i] = src frame src_idx[i] (round 5: the per-frame trunk cache of the pipelined engine) ----
// Frames are contiguous slabs in NDHWC, so "assemble the clips of a forward from cached per-frame trunk outputs" and "store the new frames'
// trunk outputs in their cache slots" are the same copy with an index table.  The table travels in the kernel arguments (<= 128 pairs: no
// device-side index buffer, nothing for a hipGraph capture to keep alive); 16-byte accesses, blocks_per_frame blocks walk one frame.
constexpr int COPY_FRAMES_MAX = 128;
struct CopyFramesParams {
    const uint4* src;
    uint4* dst;
    long long frame_vec;                 // 16-byte vectors per frame
    int n, blocks_per_frame;
    int src_idx[COPY_FRAMES_MAX], dst_idx[COPY_FRAMES_MAX];
};
__global__ __launch_bounds__(256) void copy_frames_kernel(const CopyFramesParams p) {
    const int f = blockIdx.x / p.blocks_per_frame, b = blockIdx.x - f * p.blocks_per_frame;
    const uint4* s = p.src + (size_t)p.src_idx[f] * p.frame_vec;
    uint4* d = p.dst + (size_t)p.dst_idx[f] * p.frame_vec;
    for (long long i = (long long)b * 256 + threadIdx.x; i < p.frame_vec; i += (long long)p.blocks_per_frame * 256) d[i] = s[i];
}

int dat_copy_frames(dat_ctx* ctx, dat_stream s, const void* src, const int* src_idx, void* dst, const int* dst_idx, int n,
                    long long frame_bytes) {
    DAT_ENFORCE(ctx, src && dst && src_idx && dst_idx && n >= 0 && frame_bytes > 0 && frame_bytes % 16 == 0,
                "copy_frames: bad argument (frame_bytes %lld must be a positive multiple of 16)", frame_bytes);
    for (int i0 = 0; i0 < n; i0 += COPY_FRAMES_MAX) {
        CopyFramesParams p;
        p.src = (const uint4*)src; p.dst = (uint4*)dst; p.frame_vec = frame_bytes / 16;
        p.n = n - i0 < COPY_FRAMES_MAX ? n - i0 : COPY_FRAMES_MAX;
        for (int i = 0; i < p.n; ++i) {
            DAT_ENFORCE(ctx, src_idx[i0 + i] >= 0 && dst_idx[i0 + i] >= 0, "copy_frames: negative frame index");
            p.src_idx[i] = src_idx[i0 + i]; p.dst_idx[i] = dst_idx[i0 + i];
        }
        long long bpf = (p.frame_vec + 256 * 8 - 1) / (256 * 8);      // ~8 vectors per thread
        if (bpf < 1) bpf = 1;
        if (bpf > 1024) bpf = 1024;
        p.blocks_per_frame = (int)bpf;
        hipLaunchKernelGGL(copy_frames_kernel, dim3((unsigned)(p.n * bpf)), dim3(256), 0, (hipStream_t)s, p);
    }
    DAT_CHECK_LAUNCH(ctx, "copy_frames");
    return DAT_OK;
}

int dat_time_avg(dat_ctx* ctx, dat_stream s, int dtype, const void* x, void* y, int N, int T, long long hwc) {
    DAT_ENFORCE(ctx, x && y && T > 0, "time_avg: bad argument");
    DISPATCH_DT(dtype, time_avg_kernel, dim3(grid_for((size_t)N * hwc)), dim3(TPB), (hipStream_t)s, x, y, N, T, hwc);
    DAT_CHECK_LAUNCH(ctx, "time_avg");
    return DAT_OK;
}

int dat_spatial_mean(dat_ctx* ctx, dat_stream s, int dtype, const void* x, float* y, int frames, int HW, int C, int Cs) {
    DAT_ENFORCE(ctx, x && y, "spatial_mean: null argument");
    DISPATCH_DT(dtype, spatial_mean_kernel, dim3(grid_for((size_t)frames * C)), dim3(TPB), (hipStream_t)s, x, y, frames, HW, C,
                Cs);
    DAT_CHECK_LAUNCH(ctx, "spatial_mean");
    return DAT_OK;
}

int dat_softmax_rows(dat_ctx* ctx, dat_stream s, const float* x, float* y, int rows, int K, int ld_in, int ld_out) {
    DAT_ENFORCE(ctx, x && y && K > 0, "softmax: bad argument");
    if (rows == 0) return DAT_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(grid_for(rows)), dim3(TPB), 0, (hipStream_t)s, x, y, rows, K, ld_in, ld_out);
    DAT_CHECK_LAUNCH(ctx, "softmax_rows");
    return DAT_OK;
}

int dat_deconv_k4s2_weights(dat_ctx* ctx, dat_stream s, const float* w, int Cin, int K, float* w3x3) {
    DAT_ENFORCE(ctx, w && w3x3 && Cin > 0 && K > 0, "deconv_k4s2_weights: bad argument");
    hipLaunchKernelGGL(deconv_k4s2_weights_kernel, dim3(grid_for((size_t)4 * K * Cin * 9)), dim3(TPB), 0, (hipStream_t)s, w, Cin,
                       K, w3x3);
    DAT_CHECK_LAUNCH(ctx, "deconv_k4s2_weights");
    return DAT_OK;
}

int dat_kps_finalize(dat_ctx* ctx, dat_stream s, int dtype, const void* sub, int R, int Tr, int S, int cs, int K, int up,
                     float* out) {
    DAT_ENFORCE(ctx, sub && out && up >= 2 && up % 2 == 0, "kps_finalize: up_scale must be even (detector.py:354), got %d", up);
    if (R == 0) return DAT_OK;
    const size_t total = (size_t)R * Tr * K * (2 * S * up) * (2 * S * up);
    const int M = 2 * S * up;
    const size_t lds = (size_t)(2 * S) * (2 * S) * sizeof(float);
    static const bool tile_off = getenv("DAT_KPS_FINALIZE_TILE") && atoi(getenv("DAT_KPS_FINALIZE_TILE")) == 0;   // (A/B switch)
    if (!tile_off && M <= 256 && lds <= 48 * 1024 && 4 * K <= cs && (long long)R * Tr * K < (1ll << 31) && (long long)R * Tr * K >= 1024) {   // (small jobs: the per-element kernel has more threads)
        // one block per output map: ng * M threads, ng output rows per step
        const int ng = std::max(1, std::min(4, 256 / M));
        if (dtype == DAT_BF16)
            hipLaunchKernelGGL((kps_finalize_tile_kernel<DAT_BF16>), dim3((unsigned)(R * Tr * K)), dim3(ng * M), lds, (hipStream_t)s, sub, S, cs, K, up, out);
        else
            hipLaunchKernelGGL((kps_finalize_tile_kernel<DAT_F32>), dim3((unsigned)(R * Tr * K)), dim3(ng * M), lds, (hipStream_t)s, sub, S, cs, K, up, out);
    } else {
        DISPATCH_DT(dtype, kps_finalize_kernel, dim3(grid_for(total)), dim3(TPB), (hipStream_t)s, sub, R, Tr, S, cs, K, up, out);
    }
    DAT_CHECK_LAUNCH(ctx, "kps_finalize");
    return DAT_OK;
}

int dat_split_bf16x2(dat_ctx* ctx, dat_stream s, const float* x, void* y, long long npos, int C) {
    DAT_ENFORCE(ctx, DAT_H16_FORMAT == 0, "split_bf16x2: needs the bf16 build of the library");
    DAT_ENFORCE(ctx, x && y && npos >= 0 && C > 0 && C % 64 == 0, "split_bf16x2: C %d must be a positive multiple of 64", C);
    const long long n8 = npos * (C / 8);
    if (n8 == 0) return DAT_OK;
    const int blocks = (int)std::min<long long>((n8 + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(split_bf16x2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, x, (uint4*)y, n8, C / 8);
    DAT_CHECK_LAUNCH(ctx, "split_bf16x2");
    return DAT_OK;
}

}  // extern "C"
