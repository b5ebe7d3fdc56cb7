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
};

template <int DT> __device__ __forceinline__ void mma_step(const uint4& a, const uint4& b, f32x16_t& c);
template <> __device__ __forceinline__ void mma_step<DAT_BF16>(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_step<DAT_F32>(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
}

template <int DT>
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemParams p) {
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
    // ---- weights -> LDS (16-byte pieces) ----
    for (int i = tid; i < 64 * (KPAD * ES / 16); i += 256) {
        const int row = i / (KPAD * ES / 16), pc = i - row * (KPAD * ES / 16);
        *(uint4*)(wl + row * WPITCH + pc * 16) = *(const uint4*)(p.w + ((size_t)row * KPAD * ES + pc * 16));
    }
    // ---- input patch -> LDS, channels interleaved; zero outside the frame ----
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
    for (int i = tid; i < 3 * PR * PC; i += 256) {
        const int col = i % PC;
        const int r = (i / PC) % PR;
        const int c = i / (PC * PR);
        const int ih = ih0 + r, iw = iw0 + col;
        float v = 0.f;
        if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
            v = p.data[((((size_t)n * 3 + c) * p.T + t) * p.H + ih) * p.W + iw];
        if (DT == DAT_BF16) ((uint16_t*)patch)[r * PP + col * 3 + c] = f2bf(v);
        else ((float*)patch)[r * PP + col * 3 + c] = v;
    }
    for (int i = tid; i < PR; i += 256) {                  // the two pad elements of every row: finite
        patch[i * PP + PC * 3] = 0; patch[i * PP + PC * 3 + 1] = 0;
    }
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
        sc[e] = p.scale ? p.scale[sl_c + e] : 1.f;
        bi[e] = p.bias ? p.bias[sl_c + e] : 0.f;
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

}  // extern "C"
