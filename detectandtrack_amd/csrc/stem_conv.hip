// conv1 of the ResNet3D body, fused: ConvNd [1,7,7] / stride [1,2,2] / pad [0,3,3] on the 3-channel clip + AffineChannelNd
// + ReLU (lib/modeling/ResNet3D.py:258-262), reading the reference's NC(T)HW fp32 `data` blob directly and writing the
// NDHWC activation -- no packed intermediate (the first version re-packed the clip into a 64-channel tensor, 266 MB
// written + read again, and ran a K = 256 conv for the 147 real taps).
//
// One block = 8 x 32 output positions x 64 channels.  The (2*8+6) x (2*32+6) x 3 input patch is staged in LDS with the
// channels interleaved ([row][col*3 + c], activation dtype), so the GEMM K index  k = kh*24 + (kw*3 + c)  (21 real values per
// kernel row padded to 24 with zero weights) makes every 16-byte MFMA operand piece a CONTIGUOUS run of the patch row:
// position (oh, ow), piece starting at k0 reads patch[2*oh + k0/24][6*ow + k0%24 ...].  The reads are 4-byte aligned
// (ds_read_b32 x4 / ds_read_b64 x2), 3-dword lane stride: conflict-free.  Weights [64][K] sit in LDS for the block's life.
// Output goes through the same per-wave LDS transpose as conv3d_igemm's epilogue: 16-byte channel-contiguous stores.
#include "dat_common.h"

namespace {

constexpr int TH = 8, TW = 32;                 // output tile
constexpr int PR = 2 * TH + 6, PC = 2 * TW + 6;   // patch rows / cols (one spare row + col read only under zero weights)
constexpr int PP = PC * 3 + 2;                 // patch row pitch in elements (212)

template <int DT> struct StemCfg;
template <> struct StemCfg<DAT_BF16> { static constexpr int KV = 8, KPAD = 176; };
template <> struct StemCfg<DAT_F32> { static constexpr int KV = 4, KPAD = 168; };

struct StemParams {
    const float* data;   // [N, 3, T, H, W]
    const char* w;       // [64][KPAD] activation dtype, k = kh*24 + kw*3 + c
    const float* scale;  // [64] or NULL
    const float* bias;   // [64] or NULL
    char* out;           // [N*T, Ho, Wo, 64]
    int N, T, H, W, Ho, Wo, relu;
    int tiles_h, tiles_w;
    int sb_off;          // LDS byte offset of scale[64] + bias[64] (beyond everything the phases of the kernel reuse)
};

template <int DT> __device__ __forceinline__ void mma_step(const uint4& a, const uint4& b, f32x16_t& c);
template <> __device__ __forceinline__ void mma_step<DAT_BF16>(const uint4& a, const uint4& b, f32x16_t& c) {
    c = DAT_MFMA16(a, b, c);
}
template <> __device__ __forceinline__ void mma_step<DAT_F32>(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
}

// Block prologue shared by both stem kernels: weights [64][KPAD] -> LDS in 16-byte pieces, the fp32 input patch (zero outside the
// frame) -> LDS in the activation dtype with channels interleaved, scale / bias -> LDS.
//   * Every global load is UNCONDITIONAL on a clamped address and all of them are issued before the first use: a predicated load
//     in a loop makes the compiler wait for each one (s_waitcnt vmcnt(0) per iteration), and ~25 dependent DRAM round trips were
//     the whole run time of a block (217 us for the layer; the MFMA work is ~1.4 k cycles).  (amdgpu_waves_per_eu on the kernels
//     lets the scheduler keep ~25 loads in flight instead of re-serialising them to hold 4 waves / SIMD.)
//   * One work item = one PIXEL (3 channel loads sharing the index arithmetic); only the ROWS_L x COLS_L pixels that feed kept
//     outputs are loaded, the rest of the ROWS_Z rows the MFMAs read under zero weights (k padding: kw = 7, and kh = 7 in bf16)
//     is zero-filled: it has to be finite, nothing more.
template <int DT, int ROWS_L, int COLS_L, int ROWS_Z>
__device__ __forceinline__ void stem_fill_lds(const float* __restrict__ data, const char* __restrict__ w, const float* __restrict__ scale,
                                              const float* __restrict__ bias, char* wl, typename ElemOf<DT>::type* patch, float* sb,
                                              int n, int t, int T, int H, int W, int ih0, int iw0, int tid) {
    constexpr int ES = ElemOf<DT>::size;
    constexpr int KPAD = StemCfg<DT>::KPAD;
    constexpr int WPITCH = KPAD * ES + 16;
    constexpr int PCS = KPAD * ES / 16;                    // 16-byte pieces per weight row
    constexpr int NW = 64 * PCS, WIT = (NW + 255) / 256;
    uint4 wv[WIT];
#pragma unroll
    for (int u = 0; u < WIT; ++u) {
        const int i = min(tid + u * 256, NW - 1);
        wv[u] = *(const uint4*)(w + (size_t)i * 16);       // rows are contiguous in global memory: piece i of the flat array
    }
    const float* sp = tid < 64 ? scale : bias;
    const bool has = sp != nullptr;
    const float sx = (has ? sp : data)[has ? (tid & 63) : 0];
    constexpr int ITEMS = ROWS_L * COLS_L, PIT = (ITEMS + 255) / 256;
    const float* base = data + (size_t)n * 3 * T * H * W + (size_t)t * H * W;
    const size_t cstride = (size_t)T * H * W;
    float v[PIT][3];
    int dst[PIT];
#pragma unroll
    for (int u = 0; u < PIT; ++u) {
        const int i = tid + u * 256;
        const int r = i / COLS_L, col = i - r * COLS_L;
        const int ih = ih0 + r, iw = iw0 + col;
        const bool live = (u + 1) * 256 <= ITEMS || i < ITEMS;
        const bool in = live && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        const uint32_t idx = in ? (uint32_t)(ih * W + iw) : 0u;   // uniform base + 32-bit lane offset: no 64-bit lane arithmetic
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = (base + c * cstride)[idx];
            v[u][c] = in ? x : 0.f;
        }
        dst[u] = live ? r * PP + col * 3 : -1;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < WIT; ++u) {
        const int i = min(tid + u * 256, NW - 1);          // (the tail re-writes piece NW-1 with the same value)
        const int row = i / PCS, pc = i - row * PCS;
        *(uint4*)(wl + row * WPITCH + pc * 16) = wv[u];
    }
    if (tid < 128) sb[tid] = has ? sx : (tid < 64 ? 1.f : 0.f);
#pragma unroll
    for (int u = 0; u < PIT; ++u)
        if (dst[u] >= 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (DT == DAT_BF16) ((uint16_t*)patch)[dst[u] + c] = f2bf(v[u][c]);
                else ((float*)patch)[dst[u] + c] = v[u][c];
            }
        }
    constexpr int ZT = PP - COLS_L * 3;                    // unloaded tail of a loaded row (spare columns + pitch padding)
    constexpr int NZ = ROWS_L * ZT + (ROWS_Z - ROWS_L) * PP;
    for (int i = tid; i < NZ; i += 256) {
        int at;
        if (i < ROWS_L * ZT) { const int r = i / ZT; at = r * PP + COLS_L * 3 + (i - r * ZT); }
        else at = ROWS_L * PP + (i - ROWS_L * ZT);
        patch[at] = 0;
    }
}

// ---- the same prologue fed from the UPLOADED uint8 frames (round 6) -------------------------------------------------------------------
// On the host-frame path the clip is in HBM as uint8 HWC frames of the SOURCE resolution (3 bytes per source pixel); dat_preprocess_frames
// turned them into the fp32 NC(T)HW `data` blob (resize x scale, mean subtraction, padding: 12 bytes per network pixel, 99 MB per 720p clip
// written and read again).  Here the patch loader computes the value of `data` at (c, ih, iw) ITSELF, in the arithmetic of that kernel
// (preprocess.hip; OpenCV's float32 bilinear path in the order of the host restatement): source coordinate formed in double and rounded to
// float32, horizontal pass first (columns outside the image snap onto the border pixel with weights (1, 0)), then the vertical blend (weights
// kept, row indices clamped), the mean subtracted in double and rounded once -- every product and sum rounded to float32 (contraction off
// inside the function), so the patch and therefore pool1 are BIT-IDENTICAL to dat_preprocess_frames + dat_stem_conv_pool, and the blob never
// exists.  The six bytes a pixel needs per source row (two BGR pixels) come as three aligned dwords.
struct U8Src {
    const uint8_t* frames;       // [F][h][w][3]
    int h, w;                    // source frame
    int oh, ow;                  // resized image inside the (padded) blob
    double inv_fx, inv_fy;       // 1 / scale
    double mean[3];
    long long last_dword;        // byte offset of the dword that holds the last byte of the frames buffer (loads are clamped onto it)
};

__device__ __forceinline__ void u8_src_coord(int d, double inv_scale, int n, bool snap, int* i0, int* i1, float* t) {
#pragma clang fp contract(off)
    const float f = (float)(((double)d + 0.5) * inv_scale - 0.5);
    const float fl = floorf(f);
    int s = (int)fl;
    float tt = f - fl;
    if (snap) {                     // x: out-of-range columns read the border pixel with weights (1, 0)
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

template <int DT, int ROWS_L, int COLS_L, int ROWS_Z>
__device__ __forceinline__ void stem_fill_lds_u8(const U8Src& src, const char* __restrict__ w, const float* __restrict__ scale,
                                                 const float* __restrict__ bias, char* wl, typename ElemOf<DT>::type* patch, float* sb,
                                                 int f, int H, int W, int ih0, int iw0, int tid) {
#pragma clang fp contract(off)
    constexpr int ES = ElemOf<DT>::size;
    constexpr int KPAD = StemCfg<DT>::KPAD;
    constexpr int WPITCH = KPAD * ES + 16;
    constexpr int PCS = KPAD * ES / 16;
    constexpr int NW = 64 * PCS, WIT = (NW + 255) / 256;
    uint4 wv[WIT];
#pragma unroll
    for (int u = 0; u < WIT; ++u) {
        const int i = min(tid + u * 256, NW - 1);
        wv[u] = *(const uint4*)(w + (size_t)i * 16);
    }
    const float* sp = tid < 64 ? scale : bias;
    const bool has = sp != nullptr;
    const float one = 1.f;
    const float sx = (has ? sp : &one)[has ? (tid & 63) : 0];
    constexpr int ITEMS = ROWS_L * COLS_L, PIT = (ITEMS + 255) / 256;
    const uint8_t* fr = src.frames + (size_t)f * src.h * src.w * 3;
    const long long fr_off = (long long)f * src.h * src.w * 3;
    // ---- all loads first: per pixel 2 source rows x 3 aligned dwords (unconditional, clamped addresses) ----
    uint32_t q[PIT][2][3];
#pragma unroll
    for (int u = 0; u < PIT; ++u) {
        const int i = tid + u * 256;
        const int r = i / COLS_L, col = i - r * COLS_L;
        const int y = min(max(ih0 + r, 0), src.oh - 1), x = min(max(iw0 + col, 0), src.ow - 1);
        int y0, y1, x0, x1;
        float ty, tx;
        u8_src_coord(y, src.inv_fy, src.h, false, &y0, &y1, &ty);
        u8_src_coord(x, src.inv_fx, src.w, true, &x0, &x1, &tx);
        const int xl = min(x0, src.w - 2);                  // the run starts at pixel xl: (xl, xl + 1) hold (x0, x1) -- or x0 twice at the right border
        const int yy[2] = {y0, y1};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long long o = fr_off + ((long long)yy[k] * src.w + xl) * 3;
            const long long b = o & ~3ll;
            const uint8_t* base = src.frames + b;
            q[u][k][0] = *(const uint32_t*)base;
            q[u][k][1] = *(const uint32_t*)(src.frames + min(b + 4, src.last_dword));
            q[u][k][2] = *(const uint32_t*)(src.frames + min(b + 8, src.last_dword));
        }
    }
    (void)fr;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < WIT; ++u) {
        const int i = min(tid + u * 256, NW - 1);
        const int row = i / PCS, pc = i - row * PCS;
        *(uint4*)(wl + row * WPITCH + pc * 16) = wv[u];
    }
    if (tid < 128) sb[tid] = has ? sx : (tid < 64 ? 1.f : 0.f);
#pragma unroll
    for (int u = 0; u < PIT; ++u) {
        const int i = tid + u * 256;
        const int r = i / COLS_L, col = i - r * COLS_L;
        const int ih = ih0 + r, iw = iw0 + col;
        const bool live = (u + 1) * 256 <= ITEMS || i < ITEMS;
        // inside the blob AND inside the resized image (the rest of the blob is the zero padding up to the stride multiple)
        const bool in = live && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && ih < src.oh && iw < src.ow;
        const int y = min(max(ih, 0), src.oh - 1), x = min(max(iw, 0), src.ow - 1);
        int y0, y1, x0, x1;
        float ty, tx;
        u8_src_coord(y, src.inv_fy, src.h, false, &y0, &y1, &ty);
        u8_src_coord(x, src.inv_fx, src.w, true, &x0, &x1, &tx);
        const int xl = min(x0, src.w - 2);
        const bool second = x0 != xl;                        // x0 is the SECOND pixel of the run (right border: x0 = x1 = w - 1)
        const float ux = 1.f - tx, uy = 1.f - ty;
        uint32_t lo[2], hi[2];                               // bytes 0-3 / 4-7 of the run of each row
        const int yy[2] = {y0, y1};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long long o = fr_off + ((long long)yy[k] * src.w + xl) * 3;
            const long long b = o & ~3ll;
            const unsigned sh = (unsigned)(o - b);
            // (a dword whose address was clamped is one the six bytes of the run do not reach into: both pixels of a run lie inside the frame)
            lo[k] = __builtin_amdgcn_alignbyte(q[u][k][1], q[u][k][0], sh);
            hi[k] = __builtin_amdgcn_alignbyte(q[u][k][2], q[u][k][1], sh);
        }
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t p0 = (lo[k] >> (8 * c)) & 0xffu;                                          // pixel xl, channel c
                const uint32_t p1 = c == 0 ? (lo[k] >> 24) : ((hi[k] >> (8 * (c - 1))) & 0xffu);        // pixel xl + 1
                a[k][0] = (float)((double)(second ? p1 : p0) - src.mean[c]);
                a[k][1] = (float)((double)p1 - src.mean[c]);
            }
            const float top = a[0][0] * ux + a[0][1] * tx;   // horizontal pass (two rounded products, one rounded sum)
            const float bot = a[1][0] * ux + a[1][1] * tx;
            v[c] = in ? top * uy + bot * ty : 0.f;           // vertical blend
        }
        if (live) {
            const int dst = r * PP + col * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (DT == DAT_BF16) ((uint16_t*)patch)[dst + c] = f2bf(v[c]);
                else ((float*)patch)[dst + c] = v[c];
            }
        }
    }
    constexpr int ZT = PP - COLS_L * 3;
    constexpr int NZ = ROWS_L * ZT + (ROWS_Z - ROWS_L) * PP;
    for (int i = tid; i < NZ; i += 256) {
        int at;
        if (i < ROWS_L * ZT) { const int r = i / ZT; at = r * PP + COLS_L * 3 + (i - r * ZT); }
        else at = ROWS_L * PP + (i - ROWS_L * ZT);
        patch[at] = 0;
    }
}

template <int DT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void stem_conv_kernel(const StemParams p) {
    typedef typename ElemOf<DT>::type E;
    constexpr int ES = ElemOf<DT>::size;
    constexpr int KV = StemCfg<DT>::KV, KPAD = StemCfg<DT>::KPAD;
    constexpr int WPITCH = KPAD * ES + 16;                 // weight row pitch (bytes): rows rotate through all 16-B bank quads
    constexpr int NSTEP = KPAD / (2 * KV);                 // MFMA steps (two 16-B pieces per step: k halves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;                                       // 64 x WPITCH
    E* patch = (E*)(smem + 64 * WPITCH);                   // PR x PP
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned b = blockIdx.x;
    const int tw_i = b % p.tiles_w; b /= p.tiles_w;
    const int th_i = b % p.tiles_h;
    const int f = b / p.tiles_h;                           // frame n*T + t
    const int n = f / p.T, t = f - n * p.T;
    const int oh0 = th_i * TH, ow0 = tw_i * TW;
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
    float* sb = (float*)(smem + p.sb_off);                 // scale[64], bias[64]
    stem_fill_lds<DT, PR - 1, PC - 1, PR>(p.data, p.w, p.scale, p.bias, wl, patch, sb, n, t, p.T, p.H, p.W, ih0, iw0, tid);
    __syncthreads();

    const int khalf = lane >> 5, nl = lane & 31;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
    for (int ks = 0; ks < NSTEP; ++ks) {
        const int piece = 2 * ks + khalf;                  // this lane's 16-byte K piece
        const int k0 = piece * KV;
        const int kh = k0 / 24, off = k0 - kh * 24;
        uint4 a[2], bb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *(const uint4*)(wl + (i * 32 + nl) * WPITCH + piece * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ohl = 2 * wave + j;                  // tile row of this 32-position group
            const E* src = patch + (2 * ohl + kh) * PP + 6 * nl + off;
            if (DT == DAT_BF16) {
                const uint32_t* s32 = (const uint32_t*)src;    // 4-byte aligned: (212*r + 6*n + off) is even
                bb[j] = make_uint4(s32[0], s32[1], s32[2], s32[3]);
            } else {
                const uint2* s64 = (const uint2*)src;          // 8-byte aligned
                const uint2 lo = s64[0], hi = s64[1];
                bb[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma_step<DT>(a[i], bb[j], acc[i][j]);
    }
    // ---- epilogue: per-wave LDS transpose, affine + ReLU, 16-byte channel-contiguous stores ----
    __syncthreads();                                       // patch / weights no longer needed
    constexpr int EPITCH = 64 * 4 + 16;
    constexpr int CPL = 16 / ES, LPP = 64 / CPL, PPI = 64 / LPP;
    char* est = smem + wave * (32 * EPITCH);
    const int sl_c = (lane % LPP) * CPL, sl_p = lane / LPP;
    float sc[CPL], bi[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        sc[e] = sb[sl_c + e];
        bi[e] = sb[64 + sl_c + e];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(est + nl * EPITCH + (i * 32 + g * 8 + khalf * 4) * 4) =
                    make_float4(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
        const int oh = oh0 + 2 * wave + j;
#pragma unroll
        for (int q = 0; q < 32 / PPI; ++q) {
            const int pl = q * PPI + sl_p;
            const int ow = ow0 + pl;
            float v[CPL];
#pragma unroll
            for (int e4 = 0; e4 < CPL / 4; ++e4) {
                const float4 tt = *(const float4*)(est + pl * EPITCH + (sl_c + e4 * 4) * 4);
                v[e4 * 4 + 0] = tt.x; v[e4 * 4 + 1] = tt.y; v[e4 * 4 + 2] = tt.z; v[e4 * 4 + 3] = tt.w;
            }
            if (oh >= p.Ho || ow >= p.Wo) continue;
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                v[e] = v[e] * sc[e] + bi[e];
                if (p.relu) v[e] = fmaxf(v[e], 0.f);
            }
            char* yp = p.out + ((((size_t)f * p.Ho + oh) * p.Wo + ow) * 64 + sl_c) * ES;
            if (DT == DAT_BF16) {
                *(uint4*)yp = make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]),
                                         f2bf2(v[4 % CPL], v[5 % CPL]),
                                         f2bf2(v[6 % CPL], v[7 % CPL]));
            } else {
                *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- conv1 + AffineChannelNd + ReLU + pool1 in ONE kernel ----------------------------------------------------------------------
// ResNet3D.py:258-265: ConvNd [1,7,7]/[1,2,2] -> AffineChannelNd -> Relu -> MaxPool [1,3,3]/[1,2,2] pad [0,1,1].  `conv1` is the
// largest blob of the network (64 channels at half resolution: 264 MB in bf16 for an 8 x 768 x 1344 clip) and has exactly one
// reader; the fused kernel never writes it: a block computes the 7 x 32 conv outputs under a 3 x 15 tile of pooled positions
// (rows 2*ph-1 .. 2*ph+1: one conv row / column of overlap between neighbouring blocks, +24 % conv arithmetic on a 39-GFLOP
// layer), rounds them to the activation dtype exactly as the unfused kernel stores them, stages them in LDS and writes the
// 3 x 3 maxima.  Window cells outside the conv frame are clamped onto the frame edge (a duplicate never changes a maximum), which
// is MaxPool's -inf padding.  Results are bit-identical to dat_stem_conv + dat_maxpool_hw.
constexpr int FPH = 3, FPW = 15;                 // pooled tile
constexpr int FTH = 2 * FPH + 1;                 // conv rows per block (7), one MFMA group of 32 columns each
constexpr int FPR = 2 * 8 + 6;                   // patch rows (sized for 8 groups: wave 3 computes one unused row)
constexpr int SPITCH_PAD = 8;                    // staging: bytes added to a position's channel row (bank spread of the 8-byte writes)

struct StemPoolParams {
    const float* data;
    U8Src u8;            // SRC = 1: the uploaded frames instead of `data`
    const char* w;
    const float* scale;
    const float* bias;
    char* out;           // [N*T, Hp, Wp, 64]
    int N, T, H, W, Ho, Wo, Hp, Wp, relu;
    int tiles_h, tiles_w;
    int sb_off;
};

// (occupancy window 3..4 waves per SIMD: round 4's same-box A/B at the 4-clip forward -- 2..2: 547 us, 2..3: 464-467, 3..4 / 4..4 / 4..5:
//  444-446; four blocks per CU is what the 34 KB of LDS per block allow)
template <int DT, int SRC = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void stem_pool_kernel(const StemPoolParams p) {
    typedef typename ElemOf<DT>::type E;
    constexpr int ES = ElemOf<DT>::size;
    constexpr int KV = StemCfg<DT>::KV, KPAD = StemCfg<DT>::KPAD;
    constexpr int WPITCH = KPAD * ES + 16;
    constexpr int NSTEP = KPAD / (2 * KV);
    constexpr int SPITCH = 64 * ES + SPITCH_PAD;           // staging bytes per conv position
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;                                       // 64 x WPITCH
    E* patch = (E*)(smem + 64 * WPITCH);                   // FPR x PP
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned b = blockIdx.x;
    const int tw_i = b % p.tiles_w; b /= p.tiles_w;
    const int th_i = b % p.tiles_h;
    const int f = b / p.tiles_h;
    const int n = f / p.T, t = f - n * p.T;
    const int ph0 = th_i * FPH, pw0 = tw_i * FPW;
    const int oh0 = 2 * ph0 - 1, ow0 = 2 * pw0 - 1;        // conv coordinate of the region's corner (may be -1: never read back)
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
    float* sb = (float*)__builtin_assume_aligned(smem + p.sb_off, 16);   // scale[64], bias[64] (beyond the staging area)
    if (SRC == 1)
        stem_fill_lds_u8<DT, 2 * FTH + 5, 2 * (2 * FPW + 1) + 5, 2 * FTH + 6>(p.u8, p.w, p.scale, p.bias, wl, patch, sb, f, p.H, p.W, ih0, iw0, tid);
    else
        stem_fill_lds<DT, 2 * FTH + 5, 2 * (2 * FPW + 1) + 5, 2 * FTH + 6>(p.data, p.w, p.scale, p.bias, wl, patch, sb, n, t, p.T, p.H, p.W, ih0, iw0, tid);
    __syncthreads();

    const int khalf = lane >> 5, nl = lane & 31;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* a_lane = wl + nl * WPITCH + khalf * 16;
    const E* b_lane = patch + 2 * wave * PP + 6 * nl;
#pragma unroll
    for (int ks = 0; ks < NSTEP; ++ks) {                   // fully unrolled: every LDS offset below is an immediate or one select
        constexpr int KVc = KV;
        const int k00 = 2 * ks * KVc, k01 = (2 * ks + 1) * KVc;                  // k of the two 16-byte pieces of this step
        const int o0 = (k00 / 24) * PP + k00 % 24, o1 = (k01 / 24) * PP + k01 % 24;
        const int boff = khalf ? o1 : o0;
        uint4 a[2], bb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *(const uint4*)(a_lane + i * 32 * WPITCH + ks * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) {                      // conv row (group) wave + 4*j: rows 0..6 are used, row 7 is scratch
            const E* src = b_lane + 8 * j * PP + boff;
            if (DT == DAT_BF16) {
                const uint32_t* s32 = (const uint32_t*)src;
                bb[j] = make_uint4(s32[0], s32[1], s32[2], s32[3]);
            } else {
                const uint2* s64 = (const uint2*)src;
                const uint2 lo = s64[0], hi = s64[1];
                bb[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma_step<DT>(a[i], bb[j], acc[i][j]);
    }
    __syncthreads();                                       // patch / weights no longer needed: the staging area reuses them
    // ---- affine + ReLU, rounded to the activation dtype, into the staging area [conv row][col][64 channels] ----
    char* stage = smem;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int grp = wave + 4 * j;
        if (grp >= FTH) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = i * 32 + g * 8 + khalf * 4;
                const float4 s4 = *(const float4*)(sb + c0), b4 = *(const float4*)(sb + 64 + c0);
                const float sc[4] = {s4.x, s4.y, s4.z, s4.w}, bi[4] = {b4.x, b4.y, b4.z, b4.w};
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e] * sc[e] + bi[e];
                char* dst = stage + (grp * 32 + nl) * SPITCH + c0 * ES;
                if (DT == DAT_BF16) {
                    // ReLU on the rounded values as packed signed 16-bit maxima with 0 (negative bf16 = negative int16, -0 -> +0:
                    // the pooling below orders bit patterns); rounding and ReLU commute
                    typedef short i16x2 __attribute__((ext_vector_type(2)));
                    const i16x2 lo = p.relu ? i16x2(0) : i16x2((short)-32768);
                    const uint32_t w0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, f2bf2(v[0], v[1])), lo));
                    const uint32_t w1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, f2bf2(v[2], v[3])), lo));
                    *(uint2*)dst = make_uint2(w0, w1);
                } else {
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
    }
    __syncthreads();
    // ---- 3 x 3 / stride 2 maxima: one thread = 16 bytes of channels of one pooled position ----
    constexpr int V = 16 / ES, CG = 64 / V;
    for (int it = tid; it < FPH * FPW * CG; it += 256) {
        const int cg = it % CG, pp = it / CG;
        const int ph = pp / FPW, pw = pp - ph * FPW;
        const int phg = ph0 + ph, pwg = pw0 + pw;
        if (phg >= p.Hp || pwg >= p.Wp) continue;
        int crow[3], ccol[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {                      // window cells clamped onto the frame: local conv row / column
            crow[d] = (min(max(2 * phg - 1 + d, 0), p.Ho - 1) - oh0) * 32 * SPITCH;
            ccol[d] = (min(max(2 * pwg - 1 + d, 0), p.Wo - 1) - ow0) * SPITCH + cg * 16;
        }
        uint4 o;
        if (DT == DAT_BF16 && p.relu) {
            // after ReLU every staged value is >= +0, and non-negative bf16 order like their bit patterns: packed u16 maxima
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            u16x2 m[4] = {u16x2(0), u16x2(0), u16x2(0), u16x2(0)};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const uint4 v = *(const uint4*)(stage + crow[dy] + ccol[dx]);
                    m[0] = __builtin_elementwise_max(m[0], __builtin_bit_cast(u16x2, v.x));
                    m[1] = __builtin_elementwise_max(m[1], __builtin_bit_cast(u16x2, v.y));
                    m[2] = __builtin_elementwise_max(m[2], __builtin_bit_cast(u16x2, v.z));
                    m[3] = __builtin_elementwise_max(m[3], __builtin_bit_cast(u16x2, v.w));
                }
            o = make_uint4(__builtin_bit_cast(uint32_t, m[0]), __builtin_bit_cast(uint32_t, m[1]),
                           __builtin_bit_cast(uint32_t, m[2]), __builtin_bit_cast(uint32_t, m[3]));
        } else {
            float m[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const uint4 v = *(const uint4*)(stage + crow[dy] + ccol[dx]);
                    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
                    if (DT == DAT_BF16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            m[2 * e] = fmaxf(m[2 * e], bf2f((uint16_t)(u[e] & 0xffff)));
                            m[2 * e + 1] = fmaxf(m[2 * e + 1], bf2f((uint16_t)(u[e] >> 16)));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], __uint_as_float(u[e]));
                    }
                }
            if (DT == DAT_BF16) o = make_uint4(f2bf2(m[0], m[1]), f2bf2(m[2], m[3]), f2bf2(m[4], m[5]), f2bf2(m[6], m[7]));
            else o = make_uint4(__float_as_uint(m[0]), __float_as_uint(m[1]), __float_as_uint(m[2]), __float_as_uint(m[3]));
        }
        *(uint4*)(p.out + ((((size_t)f * p.Hp + phg) * p.Wp + pwg) * 64) * ES + (size_t)cg * 16) = o;
    }
}

// conv1_w [64, 3, 1, 7, 7] fp32 -> [64][KPAD] activation dtype with k = kh*24 + kw*3 + c (zeros in the padding)
template <int DT>
__global__ void stem_conv_weights_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout) {
    constexpr int KPAD = StemCfg<DT>::KPAD;
    const int total = Cout * KPAD;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int k = i % KPAD, co = i / KPAD;
        const int kh = k / 24, rem = k - kh * 24;
        float v = 0.f;
        if (kh < 7 && rem < 21) {
            const int kw = rem / 3, c = rem - kw * 3;
            v = w[(((size_t)co * 3 + c) * 7 + kh) * 7 + kw];
        }
        ElemOf<DT>::st(out, i, v);
    }
}

}  // namespace

extern "C" {

size_t dat_stem_conv_weight_bytes(int dtype) {
    return (size_t)64 * (dtype == DAT_BF16 ? StemCfg<DAT_BF16>::KPAD : StemCfg<DAT_F32>::KPAD) * dat_esize(dtype);
}

int dat_stem_conv_pack_weights(dat_ctx* ctx, dat_stream s, int dtype, const float* conv1_w, int Cout, void* packed) {
    DAT_ENFORCE(ctx, conv1_w && packed && Cout == 64, "stem_conv: conv1 must have 64 output channels (ResNet3D.py:258), got %d", Cout);
    if (dtype == DAT_BF16) hipLaunchKernelGGL(stem_conv_weights_kernel<DAT_BF16>, dim3(64), dim3(256), 0, (hipStream_t)s, conv1_w, packed, Cout);
    else hipLaunchKernelGGL(stem_conv_weights_kernel<DAT_F32>, dim3(64), dim3(256), 0, (hipStream_t)s, conv1_w, packed, Cout);
    DAT_CHECK_LAUNCH(ctx, "stem_conv_pack_weights");
    return DAT_OK;
}

int dat_stem_conv(dat_ctx* ctx, dat_stream s, int dtype, const float* data, const void* w_packed, const float* scale,
                  const float* bias, int relu, int N, int T, int H, int W, void* out) {
    DAT_ENFORCE(ctx, data && w_packed && out, "stem_conv: null argument");
    DAT_ENFORCE(ctx, dtype == DAT_BF16 || dtype == DAT_F32, "stem_conv: bad dtype %d", dtype);
    StemParams p;
    p.data = data; p.w = (const char*)w_packed; p.scale = scale; p.bias = bias; p.out = (char*)out;
    p.N = N; p.T = T; p.H = H; p.W = W; p.relu = relu;
    p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
    p.tiles_h = (p.Ho + TH - 1) / TH; p.tiles_w = (p.Wo + TW - 1) / TW;
    const size_t es = dat_esize(dtype);
    const int kpad = dtype == DAT_BF16 ? StemCfg<DAT_BF16>::KPAD : StemCfg<DAT_F32>::KPAD;
    size_t lds = (size_t)64 * (kpad * es + 16) + (size_t)PR * PP * es;
    if (lds < 4 * 32 * (64 * 4 + 16)) lds = 4 * 32 * (64 * 4 + 16);
    lds = (lds + 15) & ~(size_t)15;
    p.sb_off = (int)lds; lds += 128 * sizeof(float);
    const long long nblocks = (long long)N * T * p.tiles_h * p.tiles_w;
    DAT_ENFORCE(ctx, nblocks > 0 && nblocks < (1ll << 31), "stem_conv: grid of %lld blocks unsupported", nblocks);
    if (dtype == DAT_BF16) {
        if (dat_ensure_lds(ctx, (const void*)stem_conv_kernel<DAT_BF16>, 64 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
        hipLaunchKernelGGL(stem_conv_kernel<DAT_BF16>, dim3((unsigned)nblocks), dim3(256), lds, (hipStream_t)s, p);
    } else {
        if (dat_ensure_lds(ctx, (const void*)stem_conv_kernel<DAT_F32>, 96 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
        hipLaunchKernelGGL(stem_conv_kernel<DAT_F32>, dim3((unsigned)nblocks), dim3(256), lds, (hipStream_t)s, p);
    }
    DAT_CHECK_LAUNCH(ctx, "stem_conv");
    return DAT_OK;
}

int dat_stem_conv_pool(dat_ctx* ctx, dat_stream s, int dtype, const float* data, const void* w_packed, const float* scale,
                       const float* bias, int relu, int N, int T, int H, int W, void* out_pool) {
    DAT_ENFORCE(ctx, data && w_packed && out_pool, "stem_conv_pool: null argument");
    DAT_ENFORCE(ctx, dtype == DAT_BF16 || dtype == DAT_F32, "stem_conv_pool: bad dtype %d", dtype);
    StemPoolParams p;
    p.data = data; p.w = (const char*)w_packed; p.scale = scale; p.bias = bias; p.out = (char*)out_pool;
    p.N = N; p.T = T; p.H = H; p.W = W; p.relu = relu;
    p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
    p.Hp = (p.Ho + 2 - 3) / 2 + 1; p.Wp = (p.Wo + 2 - 3) / 2 + 1;
    DAT_ENFORCE(ctx, p.Ho >= 1 && p.Wo >= 1 && p.Hp >= 1 && p.Wp >= 1, "stem_conv_pool: input %dx%d too small", H, W);
    p.tiles_h = (p.Hp + FPH - 1) / FPH; p.tiles_w = (p.Wp + FPW - 1) / FPW;
    const size_t es = dat_esize(dtype);
    const int kpad = dtype == DAT_BF16 ? StemCfg<DAT_BF16>::KPAD : StemCfg<DAT_F32>::KPAD;
    size_t lds = (size_t)64 * (kpad * es + 16) + (size_t)FPR * PP * es;
    const size_t stage = (size_t)FTH * 32 * (64 * es + SPITCH_PAD);
    if (lds < stage) lds = stage;
    lds = (lds + 15) & ~(size_t)15;
    p.sb_off = (int)lds; lds += 128 * sizeof(float);
    const long long nblocks = (long long)N * T * p.tiles_h * p.tiles_w;
    DAT_ENFORCE(ctx, nblocks > 0 && nblocks < (1ll << 31), "stem_conv_pool: grid of %lld blocks unsupported", nblocks);
    if (dtype == DAT_BF16) {
        if (dat_ensure_lds(ctx, (const void*)stem_pool_kernel<DAT_BF16>, 64 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
        hipLaunchKernelGGL(stem_pool_kernel<DAT_BF16>, dim3((unsigned)nblocks), dim3(256), lds, (hipStream_t)s, p);
    } else {
        if (dat_ensure_lds(ctx, (const void*)stem_pool_kernel<DAT_F32>, 96 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
        hipLaunchKernelGGL(stem_pool_kernel<DAT_F32>, dim3((unsigned)nblocks), dim3(256), lds, (hipStream_t)s, p);
    }
    DAT_CHECK_LAUNCH(ctx, "stem_conv_pool");
    return DAT_OK;
}

int dat_stem_conv_pool_u8(dat_ctx* ctx, dat_stream s, int dtype, const unsigned char* frames, int n_frames, int h, int w, double fx, double fy,
                          int out_h, int out_w, int pad_h, int pad_w, const double* pixel_means, const void* w_packed, const float* scale,
                          const float* bias, int relu, void* out_pool) {
    DAT_ENFORCE(ctx, frames && pixel_means && w_packed && out_pool, "stem_conv_pool_u8: null argument");
    DAT_ENFORCE(ctx, dtype == DAT_BF16 || dtype == DAT_F32, "stem_conv_pool_u8: bad dtype %d", dtype);
    DAT_ENFORCE(ctx, n_frames > 0 && h > 0 && w >= 2 && out_h > 0 && out_w > 0 && pad_h >= out_h && pad_w >= out_w && fx > 0 && fy > 0,
                "stem_conv_pool_u8: bad geometry %dx%d -> %dx%d (pad %dx%d)", h, w, out_h, out_w, pad_h, pad_w);
    StemPoolParams p;
    p.data = nullptr; p.w = (const char*)w_packed; p.scale = scale; p.bias = bias; p.out = (char*)out_pool;
    p.N = n_frames; p.T = 1; p.H = pad_h; p.W = pad_w; p.relu = relu;
    p.u8.frames = frames; p.u8.h = h; p.u8.w = w; p.u8.oh = out_h; p.u8.ow = out_w; p.u8.inv_fx = 1.0 / fx; p.u8.inv_fy = 1.0 / fy;
    for (int c = 0; c < 3; ++c) p.u8.mean[c] = pixel_means[c];
    const long long total = (long long)n_frames * h * w * 3;
    p.u8.last_dword = ((total + 3) & ~3ll) - 4;
    p.Ho = (p.H + 6 - 7) / 2 + 1; p.Wo = (p.W + 6 - 7) / 2 + 1;
    p.Hp = (p.Ho + 2 - 3) / 2 + 1; p.Wp = (p.Wo + 2 - 3) / 2 + 1;
    DAT_ENFORCE(ctx, p.Ho >= 1 && p.Wo >= 1 && p.Hp >= 1 && p.Wp >= 1, "stem_conv_pool_u8: input %dx%d too small", p.H, p.W);
    p.tiles_h = (p.Hp + FPH - 1) / FPH; p.tiles_w = (p.Wp + FPW - 1) / FPW;
    const size_t es = dat_esize(dtype);
    const int kpad = dtype == DAT_BF16 ? StemCfg<DAT_BF16>::KPAD : StemCfg<DAT_F32>::KPAD;
    size_t lds = (size_t)64 * (kpad * es + 16) + (size_t)FPR * PP * es;
    const size_t stage = (size_t)FTH * 32 * (64 * es + SPITCH_PAD);
    if (lds < stage) lds = stage;
    lds = (lds + 15) & ~(size_t)15;
    p.sb_off = (int)lds; lds += 128 * sizeof(float);
    const long long nblocks = (long long)n_frames * p.tiles_h * p.tiles_w;
    DAT_ENFORCE(ctx, nblocks > 0 && nblocks < (1ll << 31), "stem_conv_pool_u8: grid of %lld blocks unsupported", nblocks);
    if (dtype == DAT_BF16) {
        if (dat_ensure_lds(ctx, (const void*)stem_pool_kernel<DAT_BF16, 1>, 64 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
        hipLaunchKernelGGL((stem_pool_kernel<DAT_BF16, 1>), dim3((unsigned)nblocks), dim3(256), lds, (hipStream_t)s, p);
    } else {
        if (dat_ensure_lds(ctx, (const void*)stem_pool_kernel<DAT_F32, 1>, 96 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
        hipLaunchKernelGGL((stem_pool_kernel<DAT_F32, 1>), dim3((unsigned)nblocks), dim3(256), lds, (hipStream_t)s, p);
    }
    DAT_CHECK_LAUNCH(ctx, "stem_conv_pool_u8");
    return DAT_OK;
}

}  // extern "C"
