// Training-side kernels (gfx950): weight gradient of the fused conv (reference ops: Caffe2 ConvGradient
// reached through model.AddGradientOperators, lib/modeling/model_builder.py:908-952) and the small elementwise pieces
// of the backward pass.
//
// Weight gradient as K-contiguous GEMMs.  In NDHWC both operands of
//     G[co][ci][tap] = sum_p g[p][co] * x[p + tap][ci]
// have the reduction index p (positions) as their SLOW axis, which MFMA fragments cannot read.  Both tensors are
// therefore re-packed once per layer into channel-major rows over a common padded position grid q = (clip, t, y, x):
//     gT[co][q]           = g at (t, y, x), zero outside the output extent
//     xT[plane][ci][q]    = x at (t - pt, s*y + py - ph, s*x + px - pw), zero outside the frame
// (one plane per stride parity (py, px); stride 1 has a single plane).  A tap (kt, kh, kw) is then a pure offset
// off = (kt*Hq + kh/s)*Wq + kw/s into plane (kh%s, kw%s): G[:, :, tap] = gT @ xT[plane][:, off:]^T, a GEMM whose rows are
// K-contiguous for BOTH operands -- the same fragment layout as the forward kernel (A = rows of output channels, B = rows
// of input channels, 128-byte K slices, XOR-swizzled LDS tiles, v_mfma_f32_32x32x16_bf16 / 32x32x2_f32).  bf16 rows
// need 4-byte alignment, so a second copy of every plane shifted by one element serves the odd offsets (Wq is even).
// Split-K over q across blocks; fp32 partial tiles are accumulated into G with atomics (G is tiny: Cout*Cin*taps).
#include "dat_common.h"

#include <utility>
#include <vector>

namespace dat_conv { int ctx_num_cu(dat_ctx* ctx); }      // conv_special.hip: compute units of the device

namespace {

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

constexpr int ROWB = 128;
constexpr int NT = 256;

__device__ __forceinline__ int swz(int row, int slot) { return (row * ROWB) + (((slot ^ (row >> 1)) & 7) << 4); }

template <int DT> struct MmaT;
template <> struct MmaT<DAT_BF16> {
    static constexpr int CK = 64;
    __device__ static __forceinline__ void step(const uint4& a, const uint4& b, f32x16_t& c) {
        c = DAT_MFMA16(a, b, c);
    }
};
template <> struct MmaT<DAT_F32> {
    static constexpr int CK = 32;
    __device__ static __forceinline__ void step(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// ---- channel-major re-pack: NDHWC [clips*T, Hs, Ws, cs] -> rows [C][Qa] over the grid (clip, tf < Tq, yy < Hq, xx < Wq) ----
struct PackParams {
    const char* src;
    char* dst;            // [C][Qa]
    int clips, T, Hs, Ws, cs, C;
    int Tq, Hq, Wq;
    int st;               // spatial stride of the sampling (1 | 2)
    int ot, oy, ox;       // source coordinate = (tf - ot, st*yy + oy, st*xx + ox)
    int shift;            // element shift of this copy (0 | 1): dst[c][q] = value(q + shift)
    long long Qtot, Qa;
};

template <int DT>
__global__ __launch_bounds__(256) void cq_pack_kernel(const PackParams p) {
    typedef typename ElemOf<DT>::type E;
    constexpr int V = 16 / ElemOf<DT>::size;          // elements per 16-byte vector (8 bf16 / 4 fp32)
    constexpr int PITCH = 64 + V;                     // LDS row pitch in elements: keeps rows 16-B aligned, rotates banks
    __shared__ __attribute__((aligned(16))) E tile[64 * PITCH];
    const long long q0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    // load: 64 positions x 64 channels, one 16-byte vector of channels per thread-iteration (channel-contiguous reads)
    for (int i = threadIdx.x; i < 64 * (64 / V); i += 256) {
        const int ql = i / (64 / V), cv = (i - ql * (64 / V)) * V;
        const long long q = q0 + ql + p.shift;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q < p.Qtot && c0 + cv < p.C) {
            long long r = q;
            const int xx = (int)(r % p.Wq); r /= p.Wq;
            const int yy = (int)(r % p.Hq); r /= p.Hq;
            const int tf = (int)(r % p.Tq);
            const int n = (int)(r / p.Tq);
            const int t = tf - p.ot, y = p.st * yy + p.oy, x = p.st * xx + p.ox;
            if (t >= 0 && t < p.T && y >= 0 && y < p.Hs && x >= 0 && x < p.Ws)
                v = *(const uint4*)((const E*)p.src + (((size_t)(n * p.T + t) * p.Hs + y) * p.Ws + x) * p.cs + c0 + cv);
        }
        *(uint4*)(tile + ql * PITCH + cv) = v;        // channel strides are multiples of 4 (8 for bf16 tensors): aligned
    }
    __syncthreads();
    // store: each channel row gets 64 consecutive q, 16 bytes (V positions) per thread-iteration
    for (int i = threadIdx.x; i < 64 * (64 / V); i += 256) {
        const int cl = i / (64 / V), qv = (i - cl * (64 / V)) * V;
        if (c0 + cl >= p.C) continue;
        E e[V];
#pragma unroll
        for (int j = 0; j < V; ++j) e[j] = tile[(qv + j) * PITCH + cl];
        E* dst = (E*)p.dst + (size_t)(c0 + cl) * p.Qa + q0 + qv;
        if (q0 + qv + V <= p.Qa) {
            uint4 o;
            if constexpr (V == 8) {
                o.x = (uint32_t)e[0] | ((uint32_t)e[1] << 16); o.y = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
                o.z = (uint32_t)e[4] | ((uint32_t)e[5] << 16); o.w = (uint32_t)e[6] | ((uint32_t)e[7] << 16);
            } else {
                o.x = __float_as_uint(e[0]); o.y = __float_as_uint(e[1]); o.z = __float_as_uint(e[2]); o.w = __float_as_uint(e[3]);
            }
            *(uint4*)dst = o;
        } else {
            for (int j = 0; j < V && q0 + qv + j < p.Qa; ++j) dst[j] = e[j];
        }
    }
}

// ---- the GEMM: G[co][ci][tap] += sum_{q in [kbeg, kend)} gT[co][q] * xT[ci][q + off] -------------------------------------
struct WgradParams {
    const char* gT;       // [Cout][Qa]
    const char* xT;       // planes/copies laid out by the host: plane_stride bytes apart
    float* G;             // [Cout][Cin][ntaps] fp32
    int Cout, Cin, ntaps;
    long long Qa;         // row length (elements)
    long long K;          // number of q to reduce (multiple of CK)
    long long k_begin;    // first q of the reduction (multiple of CK): g is known to be zero before it
    int ksplit;
    int n_ci_tiles, n_co_tiles;
    long long tap_off[32];     // element offset into the B row (already reduced by the copy's shift)
    long long tap_base[32];    // byte offset of the (plane, copy) this tap reads
};

template <int DT>
__global__ __launch_bounds__(NT, 2) void wgrad_gemm_kernel(const WgradParams p) {
    constexpr int ES = ElemOf<DT>::size;
    constexpr int CK = MmaT<DT>::CK;
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 128 * ROWB];   // double buffer of (A tile, B tile)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave & 1, wave_n = wave >> 1;
    // taps fastest, then channel tiles, K-split slowest: the blocks resident together read the SAME K-chunk of the gT / xT rows
    // (a tap only shifts the xT window by a few elements), so the chunk is fetched from HBM once and re-read from L2
    unsigned bid = blockIdx.x;
    const int tap = bid % p.ntaps; bid /= p.ntaps;
    const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
    const int co_t = bid % p.n_co_tiles;
    const int split = bid / p.n_co_tiles;
    const long long ksteps = p.K / CK;
    const long long s_lo = ksteps * split / p.ksplit, s_hi = ksteps * (split + 1) / p.ksplit;

    // Both tiles stream global -> LDS by LDS-DMA into a double buffer (one barrier per K step, the tiles of step s+1 land while
    // step s computes).  One wave-instruction lands 8 rows x 128 B lane-linearly, so the XOR swizzle of the fragment reads is
    // applied on the source side: lane (row, phys slot) fetches logical slot phys ^ ((row >> 1) & 7).  The xT rows start at an
    // element offset that is only 4-byte aligned (tap offset): dword alignment is all the DMA needs.
    const int r0 = tid >> 3;                                   // row of item 0; item i: r0 + 32*i
    const int lslot = (tid & 7) ^ ((tid >> 4) & 7);            // logical slot this lane fetches ((row >> 1) & 7 is i-independent)
    const char* arow[4];
    const char* brow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = min(co_t * 128 + r0 + 32 * i, p.Cout - 1);     // clamped rows are masked at the store
        const int ci = min(ci_t * 128 + r0 + 32 * i, p.Cin - 1);
        arow[i] = p.gT + ((size_t)co * p.Qa) * ES + lslot * 16;
        brow[i] = p.xT + p.tap_base[tap] + ((size_t)ci * p.Qa + p.tap_off[tap]) * ES + lslot * 16;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#define TILE_DMA(S_, BUF_)                                                                                          \
    {                                                                                                               \
        const size_t ko_ = ((size_t)p.k_begin + (size_t)(S_) * CK) * ES;                                           \
        char* ad_ = smem + (BUF_) * (2 * 128 * ROWB) + wv * (8 * ROWB);                                             \
        char* bd_ = ad_ + 128 * ROWB;                                                                               \
        _Pragma("unroll")                                                                                           \
        for (int i_ = 0; i_ < 4; ++i_) {                                                                            \
            __builtin_amdgcn_global_load_lds((gptr_t)(arow[i_] + ko_), (lptr_t)(ad_ + i_ * 32 * ROWB), 16, 0, 0);   \
            __builtin_amdgcn_global_load_lds((gptr_t)(brow[i_] + ko_), (lptr_t)(bd_ + i_ * 32 * ROWB), 16, 0, 0);   \
        }                                                                                                           \
    }

    const int khalf = lane >> 5;
    int a_off[2][4], b_off[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a_off[i][ks] = swz(wave_m * 64 + i * 32 + (lane & 31), ks * 2 + khalf);
            b_off[i][ks] = 128 * ROWB + swz(wave_n * 64 + i * 32 + (lane & 31), ks * 2 + khalf);
        }
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (s_lo < s_hi) TILE_DMA(s_lo, 0);
    for (long long s = s_lo; s < s_hi; ++s) {
        const int buf = (int)((s - s_lo) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of step s landed ...
        __syncthreads();                                      // ... and everybody's; previous step's fragment reads are done
        if (s + 1 < s_hi) TILE_DMA(s + 1, buf ^ 1);
        const char* tb = smem + buf * (2 * 128 * ROWB);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { a[i] = *(const uint4*)(tb + a_off[i][ks]); b[i] = *(const uint4*)(tb + b_off[i][ks]); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) MmaT<DT>::step(a[i], b[j], acc[i][j]);
        }
    }
#undef TILE_DMA
    // D[i = co][j = ci]: lane holds column ci = lane&31; register r -> row (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Partial tiles go to Gt[tap][co][ci] (ci contiguous: a wave's 64 atomics land in 2 lines instead of 64); the finish kernel
    // transposes to the reference weight layout.  With a single K split the tile is complete: plain stores.
    float* Gt = p.G + (size_t)tap * p.Cout * p.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci_t * 128 + wave_n * 64 + j * 32 + (lane & 31);
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_t * 128 + wave_m * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (co >= p.Cout) continue;
                if (p.ksplit > 1) atomicAdd(Gt + (size_t)co * p.Cin + ci, acc[i][j][r]);
                else Gt[(size_t)co * p.Cin + ci] = acc[i][j][r];
            }
        }
}

// ---- the direct weight-gradient GEMM (round 3, bf16): no re-pack passes ------------------------------------------------------------
// G[co][ci][tap] = sum_p g[p][co] * x[in(p, tap)][ci] reduces over POSITIONS, the slow axis of both NDHWC tensors, while an MFMA
// fragment wants 8 consecutive k per lane.  Rounds 1-2 re-packed both tensors channel-major first (cq_pack_kernel: 2-5 extra passes
// per layer, 7-10 % of a training iteration).  gfx950 has a transposing LDS read, ds_read_b64_tr_b16: within a 16-lane group lane i
// supplies the address of 4 contiguous bf16 (row i / 4, columns 4 (i % 4) ... of a 4 x 16 block, any row pitch) and receives
// column i of the block -- 4 consecutive ROWS (probed on the box: tools/probes/tr_probe.hip).  So the tiles are staged in LDS exactly as
// they lie in memory -- [64 positions][128 channels], one 256-byte run per position -- and read k-contiguous by the hardware:
//   * a block owns a (128 co x 128 ci) tile of one tap and a K range; per chunk of 64 output positions it stages the g rows and the x
//     rows of those positions' tap inputs (zero rows outside the frame: the conv's padding; stride folded into the row address);
//   * global -> registers -> LDS, the next chunk's loads in flight behind the current chunk's MFMAs, double-buffered LDS, one barrier
//     per chunk; row pitch 320 B (+64): a group's 4 rows x 8 dwords and the neighbouring group's columns fall on distinct banks;
//   * per 16-k step a wave (64 x 64) issues 8 transposing reads for 4 v_mfma_f32_32x32x16_bf16.
// Partial tiles go to Gt[tap][co][ci] exactly like wgrad_gemm_kernel (atomics when the K range is split), wgrad_finish_kernel follows.
struct WgradDirectParams {
    const char* g;        // [frames * Ho * Wo][g_cs] bf16
    const char* x;        // [frames * H * W][x_cs] bf16
    float* G;             // [ntaps][Cout][Cin] fp32
    int Cout, Cin, g_cs, x_cs;
    int T, H, W, Ho, Wo, stride;
    int KT, KH, KW, pt, ph, pw;
    int ntaps, ksplit, n_ci_tiles, n_co_tiles;
    int atomic;                 // add into G with atomics (K split, or a caller-owned accumulator: dat_conv3d_wgrad_acc) instead of storing
    unsigned p_begin, p_end;    // output positions [p_begin, p_end) carry a non-zero gradient (frame window)
    unsigned how, wo_magic;     // Ho * Wo; ceil(2^32 / Wo): row = umulhi(position in frame, wo_magic) (Ho * Wo * Wo < 2^32 checked by the launcher)
};

constexpr int WD_PITCH = 320;                 // bytes per staged position row: 128 channels + 64 B (bank rotation, see above)
constexpr int WD_TILE = 64 * WD_PITCH;        // one operand tile: 64 positions

// One MFMA fragment = two transposing reads (k 0..3 | 4..7 of the lane's fragment row: rows +4 of the staged tile).  The compiler
// builtin (not inline asm): it knows these are LDS loads, counts lgkmcnt for them and schedules them ahead of the MFMAs -- an asm
// version with its own s_waitcnt could not be pipelined (the hardware does not interlock VGPR reads on outstanding LDS operations,
// so nothing may touch an asm read's result before a wait the compiler cannot see).
typedef short wd_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 wd_tr_frag(const char* base) {
    typedef __attribute__((address_space(3))) wd_v4s* lp_t;
    const wd_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(base));
    const wd_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(base + 4 * WD_PITCH));
    uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}

__global__ __launch_bounds__(NT, 2) void wgrad_direct_kernel(const WgradDirectParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages x (g tile, x tile)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave & 1, wave_n = wave >> 1;
    unsigned bid = blockIdx.x;
    const int tap = bid % p.ntaps; bid /= p.ntaps;
    const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
    const int co_t = bid % p.n_co_tiles;
    const int split = bid / p.n_co_tiles;
    const int kw = tap % p.KW, kh = (tap / p.KW) % p.KH, kt = tap / (p.KW * p.KH);
    const unsigned nchunks = (p.p_end - p.p_begin + 63u) / 64u;
    const unsigned c_lo = (unsigned)((unsigned long long)nchunks * split / p.ksplit), c_hi = (unsigned)((unsigned long long)nchunks * (split + 1) / p.ksplit);

    // staging: thread -> 16-byte piece `pc` (0..15) of rows r0 + 16 i (i < 4) of both tiles
    const int pc = tid & 15, r0 = tid >> 4;
    const bool g_col_ok = co_t * 128 + pc * 8 < p.g_cs, x_col_ok = ci_t * 128 + pc * 8 < p.x_cs;
    uint4 gq[4], xq[4];
#define WD_FETCH(CH_)                                                                                                   \
    {                                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                              \
            const unsigned pos_ = p.p_begin + (unsigned)(CH_) * 64u + (unsigned)(r0 + 16 * i_);                          \
            gq[i_] = make_uint4(0, 0, 0, 0);                                                                            \
            xq[i_] = make_uint4(0, 0, 0, 0);                                                                            \
            if (pos_ < p.p_end) {                                                                                       \
                if (g_col_ok) gq[i_] = *(const uint4*)(p.g + ((size_t)pos_ * (unsigned)p.g_cs + (unsigned)(co_t * 128 + pc * 8)) * 2u); \
                const unsigned f_ = pos_ / p.how, rem_ = pos_ - f_ * p.how;                                             \
                const unsigned oy_ = __umulhi(rem_, p.wo_magic), ox_ = rem_ - oy_ * (unsigned)p.Wo;                     \
                const unsigned clip_ = f_ / (unsigned)p.T, t_ = f_ - clip_ * (unsigned)p.T;                             \
                const int ti_ = (int)t_ + kt - p.pt, yi_ = (int)oy_ * p.stride + kh - p.ph, xi_ = (int)ox_ * p.stride + kw - p.pw; \
                if (x_col_ok && ti_ >= 0 && ti_ < p.T && yi_ >= 0 && yi_ < p.H && xi_ >= 0 && xi_ < p.W) {             \
                    const size_t row_ = ((size_t)(clip_ * (unsigned)p.T + (unsigned)ti_) * (unsigned)p.H + (unsigned)yi_) * (unsigned)p.W + (unsigned)xi_; \
                    xq[i_] = *(const uint4*)(p.x + (row_ * (unsigned)p.x_cs + (unsigned)(ci_t * 128 + pc * 8)) * 2u);  \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
    }
#define WD_STAGE(BUF_)                                                                                                  \
    {                                                                                                                   \
        char* gd_ = smem + (BUF_) * (2 * WD_TILE);                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                              \
            *(uint4*)(gd_ + (r0 + 16 * i_) * WD_PITCH + pc * 16) = gq[i_];                                              \
            *(uint4*)(gd_ + WD_TILE + (r0 + 16 * i_) * WD_PITCH + pc * 16) = xq[i_];                                    \
        }                                                                                                               \
    }
    // fragment addressing: 16-lane group G = lane >> 4 reads the 4 x 16 block (rows k0 + (G >> 1) * 8 [+4], columns c0 + (G & 1) * 16);
    // lane i of the group supplies row i >> 2, columns 4 (i & 3)
    const int grp = lane >> 4, li = lane & 15;
    const int frag_off = ((grp >> 1) * 8 + (li >> 2)) * WD_PITCH + ((grp & 1) * 16 + (li & 3) * 4) * 2;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_lo < c_hi) {
        WD_FETCH(c_lo);
        WD_STAGE(0);
    }
    __syncthreads();
    for (unsigned c = c_lo; c < c_hi; ++c) {
        const int buf = (int)((c - c_lo) & 1u);
        if (c + 1 < c_hi) WD_FETCH(c + 1);          // in flight behind this chunk's MFMAs
        const char* gb = smem + buf * (2 * WD_TILE) + frag_off + (wave_m * 64) * 2;
        const char* xb = smem + buf * (2 * WD_TILE) + WD_TILE + frag_off + (wave_n * 64) * 2;
        // fragments of step ks + 1 are requested before the MFMAs of step ks (register double buffer)
        uint4 a[2][2], b[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[0][i] = wd_tr_frag(gb + i * 64);
            b[0][i] = wd_tr_frag(xb + i * 64);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[(ks + 1) & 1][i] = wd_tr_frag(gb + (ks + 1) * 16 * WD_PITCH + i * 64);
                    b[(ks + 1) & 1][i] = wd_tr_frag(xb + (ks + 1) * 16 * WD_PITCH + i * 64);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks these reads behind the MFMAs, next to their use)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) MmaT<DAT_BF16>::step(a[ks & 1][i], b[ks & 1][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < c_hi) WD_STAGE(buf ^ 1);
        __syncthreads();
    }
#undef WD_FETCH
#undef WD_STAGE
    const int khalf = lane >> 5;
    float* Gt = p.G + (size_t)tap * p.Cout * p.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci_t * 128 + wave_n * 64 + j * 32 + (lane & 31);
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_t * 128 + wave_m * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (co >= p.Cout) continue;
                if (p.atomic) atomicAdd(Gt + (size_t)co * p.Cin + ci, acc[i][j][r]);
                else Gt[(size_t)co * p.Cin + ci] = acc[i][j][r];
            }
        }
}

// ---- direct weight gradient of 3 x 3 (x KT) stride-1 convs: the nine spatial taps of one temporal tap share ONE staged input patch -------
// wgrad_direct_kernel runs one block per tap, so a 3 x 3 x 3 layer streams both tensors 27 times through L2 -> LDS (res3-res5, the FPN
// post-hoc convs and the keypoint head: ~25 GB per training iteration -- the bound of rounds 1-2's kernel too).  With positions as LDS
// ROWS a spatial tap is a row offset: a block stages the g rows of an 8 x 8 patch of output positions (64 k) and the 10 x 10 patch of
// input positions around it ONCE and accumulates all nine (kh, kw) taps from it -- the B fragment of tap (kh, kw) is the same
// per-lane address + (kh * 10 + kw) rows, an immediate.  Block = (64 co x 64 ci) x one temporal tap, a wave = 32 x 32 x 9 taps (144
// accumulator registers); K = frames x patches, split across blocks.  Row pitch 192 B (64 channels + 64 B): the four rows of a
// transposing read and the neighbouring 16-column group fall on distinct banks.  Operand traffic per layer: 9 x less.
struct Wgrad9Params {
    const char* g;
    const char* x;
    float* G;             // [ntaps][Cout][Cin] fp32
    int Cout, Cin, g_cs, x_cs;
    int T, H, W;          // stride 1, pad 1: the output map is H x W too
    int KT, pt;
    int ksplit, n_ci_tiles, n_co_tiles;
    int f_begin, f_end;   // output frames [f_begin, f_end) carry a non-zero gradient (frame window; whole clips otherwise)
    int tiles_h, tiles_w; // 8 x 8 patches per frame
    int ablate;           // DEBUG (DAT_WGRAD_ABLATE): 1 no global loads, 2 no LDS fragment reads / MFMAs, 4 no final atomics / stores
    const char* zeros;    // >= 16 zero bytes in global memory (LDS-DMA source of halo / out-of-range pieces)
    int atomic;           // add into G with atomics (K split, or a caller-owned accumulator) instead of storing
    int xcd;              // XCD-aware block order (DAT_WGRAD_XCD): consecutive logical blocks share an XCD's L2
};

constexpr int W9_PITCH = 192;
constexpr int W9_GT = 64 * W9_PITCH;          // g tile: 64 positions
constexpr int W9_XT = 100 * W9_PITCH;         // x patch: 10 x 10 positions
constexpr int W9_STAGE = W9_GT + W9_XT + 512; // (+ pad: the 12-k row reads of the last patch rows reach two rows past the patch)

__device__ __forceinline__ uint4 w9_tr_frag(const char* base) {
    typedef __attribute__((address_space(3))) wd_v4s* lp_t;
    const wd_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(base));
    const wd_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(base + 4 * W9_PITCH));
    uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}
// one transposing read: 4 consecutive k (patch rows) of this lane's column
__device__ __forceinline__ uint2 w9_tr4(const char* base) {
    typedef __attribute__((address_space(3))) wd_v4s* lp_t;
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(base)));
}

__global__ __launch_bounds__(NT, 2) void wgrad_direct9_kernel(const Wgrad9Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages x (g tile, x patch)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave & 1, wave_n = wave >> 1;
    unsigned bid = blockIdx.x;
    const int kt = bid % p.KT; bid /= p.KT;
    const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
    const int co_t = bid % p.n_co_tiles;
    const int split = bid / p.n_co_tiles;
    const unsigned per_frame = (unsigned)(p.tiles_h * p.tiles_w);
    const unsigned nchunks = (unsigned)(p.f_end - p.f_begin) * per_frame;
    const unsigned c_lo = (unsigned)((unsigned long long)nchunks * split / p.ksplit), c_hi = (unsigned)((unsigned long long)nchunks * (split + 1) / p.ksplit);

    // staging: 16-byte piece pc (8 per 64-channel row); g: rows r0 + 32 i (i < 2), x: patch rows r0 + 32 i (i < 4, < 100)
    const int pc = tid & 7, r0 = tid >> 3;
    const bool g_col_ok = co_t * 64 + pc * 8 < p.g_cs, x_col_ok = ci_t * 64 + pc * 8 < p.x_cs;
    uint4 gqA[2], xqA[4], gqB[2], xqB[4];        // two register sets: the loads of chunk c + 2 are issued while chunk c computes
#define W9_FETCH(CH_, GQ_, XQ_)                                                                                                   \
    {                                                                                                                   \
        const unsigned fr_ = (unsigned)(CH_) / per_frame, tl_ = (unsigned)(CH_) - fr_ * per_frame;                       \
        const int f_ = p.f_begin + (int)fr_;                                                                            \
        const int ty_ = (int)(tl_ / (unsigned)p.tiles_w), tx_ = (int)tl_ - ty_ * p.tiles_w;                             \
        const int clip_ = f_ / p.T, t_ = f_ - clip_ * p.T, ti_ = t_ + kt - p.pt;                                        \
        const bool tin_ = ti_ >= 0 && ti_ < p.T;                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                              \
            const int r_ = r0 + 32 * i_, oy_ = ty_ * 8 + (r_ >> 3), ox_ = tx_ * 8 + (r_ & 7);                           \
            GQ_[i_] = make_uint4(0, 0, 0, 0);                                                                            \
            if (g_col_ok && tin_ && oy_ < p.H && ox_ < p.W)                                                             \
                GQ_[i_] = *(const uint4*)(p.g + ((((size_t)f_ * p.H + oy_) * p.W + ox_) * (unsigned)p.g_cs + (unsigned)(co_t * 64 + pc * 8)) * 2u); \
        }                                                                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                              \
            const int r_ = r0 + 32 * i_, py_ = r_ / 10, px_ = r_ - py_ * 10;                                            \
            const int iy_ = ty_ * 8 + py_ - 1, ix_ = tx_ * 8 + px_ - 1;                                                 \
            XQ_[i_] = make_uint4(0, 0, 0, 0);                                                                            \
            if (x_col_ok && tin_ && r_ < 100 && iy_ >= 0 && iy_ < p.H && ix_ >= 0 && ix_ < p.W)                         \
                XQ_[i_] = *(const uint4*)(p.x + ((((size_t)(clip_ * p.T + ti_) * p.H + iy_) * p.W + ix_) * (unsigned)p.x_cs + (unsigned)(ci_t * 64 + pc * 8)) * 2u); \
        }                                                                                                               \
    }
#define W9_STAGE_WRITE(BUF_, GQ_, XQ_)                                                                                            \
    {                                                                                                                   \
        char* gd_ = smem + (BUF_) * W9_STAGE;                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) *(uint4*)(gd_ + (r0 + 32 * i_) * W9_PITCH + pc * 16) = GQ_[i_];  \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                \
            if (r0 + 32 * i_ < 100) *(uint4*)(gd_ + W9_GT + (r0 + 32 * i_) * W9_PITCH + pc * 16) = XQ_[i_];             \
    }
    // fragment addressing.  g tile (rows = k = oy * 8 + ox): group G = lane >> 4 reads rows k0 + (G >> 1) * 8 [+4], columns (G & 1) * 16,
    // lane i of the group row i >> 2, columns 4 (i & 3).  x patch: k -> patch row (oy + kh) * 10 + ox + kw; the 8 k of a fragment
    // half are ONE output row (oy = 2 ks + (G >> 1)), so its two 4-k groups are patch rows ... + ox (0..3 | 4..7): +4 rows again.
    const int grp = lane >> 4, li = lane & 15;
    const int g_off = ((grp >> 1) * 8 + (li >> 2)) * W9_PITCH + ((grp & 1) * 16 + (li & 3) * 4) * 2 + (wave_m * 32) * 2;
    const int x_off = ((grp >> 1) * 10 + (li >> 2)) * W9_PITCH + ((grp & 1) * 16 + (li & 3) * 4) * 2 + (wave_n * 32) * 2;
    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* gb = smem + buf * W9_STAGE + g_off;
        const char* xb = smem + buf * W9_STAGE + W9_GT + x_off;
        if (p.ablate & 2) return;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {            // 16 k = output rows 2 ks, 2 ks + 1 of the patch
            const uint4 a = w9_tr_frag(gb + ks * 16 * W9_PITCH);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                // the three kw taps read three 8-wide windows of the SAME 10-wide patch row: 12 k (three 4-k transposing reads; the last
                // two are past the row, never used) are fetched once and the windows cut out of the registers (LDS reads per MFMA 2.2 -> 1.2)
                const char* rb = xb + ((2 * ks + kh) * 10) * W9_PITCH;
                const uint2 q0 = w9_tr4(rb), q1 = w9_tr4(rb + 4 * W9_PITCH), q2 = w9_tr4(rb + 8 * W9_PITCH);
                const uint4 b0 = make_uint4(q0.x, q0.y, q1.x, q1.y);
                const uint4 b2 = make_uint4(q0.y, q1.x, q1.y, q2.x);
                const uint4 b1 = make_uint4(__builtin_amdgcn_alignbyte(q0.y, q0.x, 2), __builtin_amdgcn_alignbyte(q1.x, q0.y, 2),
                                            __builtin_amdgcn_alignbyte(q1.y, q1.x, 2), __builtin_amdgcn_alignbyte(q2.x, q1.y, 2));
                MmaT<DAT_BF16>::step(a, b0, acc[kh * 3 + 0]);
                MmaT<DAT_BF16>::step(a, b1, acc[kh * 3 + 1]);
                MmaT<DAT_BF16>::step(a, b2, acc[kh * 3 + 2]);
            }
        }
    };
    // Prefetch distance TWO chunks (ablation, tools/probes/wgrad_bench.py ablate: loads, MFMAs and the final atomics each cost about a
    // quarter of a launch and did not overlap -- a chunk's MFMA phase (~0.5 us) is shorter than a memory round trip): chunk c computes
    // from LDS, chunk c + 1 is in registers on its way to the other LDS buffer, chunk c + 2 is being requested.
    const bool ld = !(p.ablate & 1);
    if (c_lo < c_hi) {
        W9_FETCH(c_lo, gqA, xqA);
        W9_STAGE_WRITE(0, gqA, xqA);
        if (c_lo + 1 < c_hi && ld) W9_FETCH(c_lo + 1, gqA, xqA);
    }
    __syncthreads();
    for (unsigned c = c_lo; c < c_hi; c += 2) {
        if (c + 2 < c_hi && ld) W9_FETCH(c + 2, gqB, xqB);
        compute(0);
        if (c + 1 < c_hi) W9_STAGE_WRITE(1, gqA, xqA);
        __syncthreads();
        if (c + 1 >= c_hi) break;
        if (c + 3 < c_hi && ld) W9_FETCH(c + 3, gqA, xqA);
        compute(1);
        if (c + 2 < c_hi) W9_STAGE_WRITE(0, gqB, xqB);
        __syncthreads();
    }
#undef W9_FETCH
#undef W9_STAGE_WRITE
    const int khalf = lane >> 5;
    const int ci = ci_t * 64 + wave_n * 32 + (lane & 31);
    if (ci >= p.Cin || (p.ablate & 4)) return;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* Gt = p.G + (size_t)(kt * 9 + t) * p.Cout * p.Cin;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_t * 64 + wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (co >= p.Cout) continue;
            if (p.atomic) atomicAdd(Gt + (size_t)co * p.Cin + ci, acc[t][r]);
            else Gt[(size_t)co * p.Cin + ci] = acc[t][r];
        }
    }
}

// ---- the same tile with the operands brought in by LDS-DMA (round 3, late) -------------------------------------------------------
// wgrad_direct9_kernel stages its operands through registers: two chunks in flight cost 48 registers next to 144 accumulators, and the
// counters (profiles/r03/train_r18/pmc_mfma.csv) show the MFMA pipe busy 21 % of the time: a 64-position chunk is ~0.5 us of MFMAs
// behind a 20.8-KB operand fetch with ~2 us of loaded latency.  Here `global_load_lds_dwordx4` writes the 128-byte rows straight into
// one of THREE LDS stages (no staging registers, no ds_write pass): chunk c computes while chunks c + 1 and c + 2 are in flight.
//   * stage = 192 rows x 128 B (64 g rows, 100 patch rows, 28 spare rows that make 24 one-KiB DMA pieces: every wave issues exactly
//     six per chunk, so `s_waitcnt vmcnt(6)` at the top of an iteration means "my pieces of THIS chunk have landed");
//   * rows are contiguous (the DMA writes lane * 16 B), so the bank spread the 192-byte pitch gave the transposing reads comes from a
//     swizzle instead: 16-byte piece p of row r sits in slot p ^ 4 * bit1(r).  A transposing read touches four consecutive rows x 64 B per
//     half wave: rows r .. r + 3 then start at banks 0 / 32 / 16 / 48 -- all 64 banks once.  The swizzle is applied by choosing which
//     global piece a lane fetches; on the read side it is a per-lane constant XOR of 64 (g tile), toggled by kh == 1 for the patch rows
//     ((2 ks + kh) * 10 flips bit 1 of the row exactly when kh == 1).
constexpr int W9D_STAGE = 192 * 128;
constexpr int W9D_STAGES = 3;

// SUB = 2 (round 4): one block of EIGHT waves per CU instead of two blocks of four -- two sub-blocks with their own K range and their own three
// stages work on the same (co, ci, kt) tile and add their accumulators through LDS before the atomics.  The grid is then one block per CU (the
// ~384 four-wave blocks left half of the CUs with one block and half with two) and a tile's partial sums reach memory from half as many blocks.
// ILV = 1 (round 4, the default): the six LDS-DMA pieces of chunk c + 2 are issued BETWEEN the MFMA groups of chunk c (one piece before every other
// group of three MFMAs) instead of in one burst behind the barrier: a piece costs its wave 60-185 cycles of issue (MI355X_MICROARCH.md), and
// behind the barrier every wave of the CU pays them at the same time with no MFMA in flight.  Same box, per layer: res3 0.152 -> 0.147 ms,
// res4 0.163 -> 0.161, res5 0.221 -> 0.216, P2 0.284 -> 0.274; letting the two sub-blocks take turns (one issues in the first half of the chunk,
// the other in the second) needs a branch per group, which cuts the chunk's MFMA schedule into basic blocks: 1-2 % slower than the burst.
template <int SUB, int ILV>
__global__ __launch_bounds__(NT * SUB, SUB == 1 ? 2 : 1) void wgrad_dma9_kernel(const Wgrad9Params p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = SUB == 1 ? 0 : wave_all >> 2, wave = wave_all & 3;
    const int wave_m = wave & 1, wave_n = wave >> 1;
    unsigned bid = blockIdx.x;
    if (p.xcd) {
        // Blocks go to the XCDs round-robin (block i -> XCD i % 8), and the blocks that read the same operand chunks at the same time are the
        // (kt, ci tile) / (kt, co tile) neighbours of one K range: consecutive ids.  Unmapped, the 12 readers of a g chunk (4 ci tiles x 3 kt) sit
        // on 8 different L2s and the chunk crosses the fabric ~7 times; mapped (the bijection of conv3d_igemm_kernel), XCD x runs the
        // consecutive logical ids [x * n / 8, (x + 1) * n / 8).
        const unsigned nx = 8, q = gridDim.x / nx, r = gridDim.x % nx;
        const unsigned xcd = bid % nx, k = bid / nx;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int kt = bid % p.KT; bid /= p.KT;
    const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
    const int co_t = bid % p.n_co_tiles;
    const int split0 = (bid / p.n_co_tiles) * SUB, split = split0 + sub;
    const unsigned per_frame = (unsigned)(p.tiles_h * p.tiles_w);
    const unsigned nchunks = (unsigned)(p.f_end - p.f_begin) * per_frame;
    const unsigned c_lo = (unsigned)((unsigned long long)nchunks * split / p.ksplit), c_hi = (unsigned)((unsigned long long)nchunks * (split + 1) / p.ksplit);
    // every wave of the block passes the same barriers: iterate over the longer of the sub-blocks' ranges (chunks past a range are fetched as zeros)
    unsigned n_iter = c_hi - c_lo;
    if (SUB == 2) {
        const unsigned o_lo = (unsigned)((unsigned long long)nchunks * (split ^ 1) / p.ksplit), o_hi = (unsigned)((unsigned long long)nchunks * ((split ^ 1) + 1) / p.ksplit);
        n_iter = max(n_iter, o_hi - o_lo);
    }
    char* const sm = smem + sub * (W9D_STAGES * W9D_STAGE);

    // ---- this lane's six DMA pieces: stage row (wave + 4 u) * 8 + lane / 8, slot lane % 8 holds data piece slot ^ 4 * bit1(row) ----
    // kind 0: g row (dy, dx) = (row / 8, row % 8); kind 1: patch row (dy, dx) = (xr / 10 - 1, xr % 10 - 1); kind 2: spare (zeros)
    int pk_dy[6], pk_dx[6], pk_col[6];     // pk_col: byte offset of the piece inside the 64-channel run, -1 = always zeros
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int row = (wave + 4 * u) * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ (((row >> 1) & 1) << 2);
        int dy = 0, dx = 0, col = -1;
        if (row < 64) {
            dy = row >> 3; dx = row & 7;
            if (co_t * 64 + piece * 8 < p.g_cs) col = (co_t * 64 + piece * 8) * 2;
        } else if (row < 164) {
            const int xr = row - 64, py = xr / 10;
            dy = py - 1; dx = xr - py * 10 - 1;
            if (ci_t * 64 + piece * 8 < p.x_cs) col = (ci_t * 64 + piece * 8) * 2;
        }
        pk_dy[u] = dy; pk_dx[u] = dx; pk_col[u] = col;
    }
    const char* const zeros = p.zeros;
    struct ChunkAt { bool tin; size_t g_frame, x_frame; int ty, tx; char* dst0; };
    auto chunk_at = [&](unsigned ch, int stage) __attribute__((always_inline)) {
        // (chunks past the block's range are requested as zeros: the instruction count per chunk stays six, the counted wait stays valid)
        const bool live = ch < c_hi && !(p.ablate & 1);
        const unsigned chc = live ? ch : c_lo;
        const unsigned fr = chc / per_frame, tl = chc - fr * per_frame;
        const int f = p.f_begin + (int)fr;
        ChunkAt a;
        a.ty = (int)(tl / (unsigned)p.tiles_w); a.tx = (int)tl - a.ty * p.tiles_w;
        const int clip = f / p.T, t = f - clip * p.T, ti = t + kt - p.pt;
        a.tin = live && ti >= 0 && ti < p.T;
        a.g_frame = (size_t)f * p.H; a.x_frame = (size_t)(clip * p.T + (a.tin ? ti : 0)) * p.H;
        a.dst0 = sm + stage * W9D_STAGE + wave * 1024;
        return a;
    };
    auto issue_piece = [&](const ChunkAt& a, auto u_c) __attribute__((always_inline)) {
        constexpr int u = decltype(u_c)::value;
        const bool is_g = wave + 4 * u < 8;                         // pieces 0..7 are the g rows (uniform per wave and u)
        const int y = a.ty * 8 + pk_dy[u], x = a.tx * 8 + pk_dx[u];
        const bool ok = a.tin && pk_col[u] >= 0 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        // branch-free: the address of a clamped pixel is formed unconditionally and blended with the zero line by a mask (an `if` here
        // became a branch around every piece, which cut the MFMA schedule of the chunk into basic blocks)
        const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.W - 1), cc = max(pk_col[u], 0);
        const char* at = is_g ? p.g + (((a.g_frame + yc) * p.W + xc) * (size_t)p.g_cs) * 2 + cc
                              : p.x + (((a.x_frame + yc) * p.W + xc) * (size_t)p.x_cs) * 2 + cc;
        const unsigned long long m = ok ? ~0ull : 0ull;
        const char* src = (const char*)((unsigned long long)zeros + (((unsigned long long)at - (unsigned long long)zeros) & m));
        // (inline asm, not __builtin_amdgcn_global_load_lds: the compiler orders every later ds_read behind an LDS-DMA it can see with
        //  s_waitcnt vmcnt(0) -- one LDS array, everything may alias -- which would wait for the chunks still in flight; the counted
        //  wait + barrier at the top of the loop is the ordering this pipeline needs)
        // (M0 is written without telling the compiler: nothing else in this kernel uses it -- no other LDS-DMA, no s_movrel)
        const unsigned lds_at = (unsigned)(size_t)(lptr_t)(a.dst0 + u * 4096);
        if (ILV)    // no memory clobber: the fragment reads of the chunk being computed may be scheduled across the piece (another stage)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_at));
        else
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_at) : "memory");
    };
    auto issue = [&](unsigned ch, int stage) __attribute__((always_inline)) {
        const ChunkAt a = chunk_at(ch, stage);
        static_for(std::make_integer_sequence<int, 6>{}, [&](auto u_c) __attribute__((always_inline)) { issue_piece(a, u_c); });
    };
    // fragment addressing (see wgrad_direct9_kernel), rows 128 B apart, the piece swizzle as a per-lane XOR
    const int grp = lane >> 4, li = lane & 15;
    const int g_row = (grp >> 1) * 8 + (li >> 2), x_row = (grp >> 1) * 10 + (li >> 2);
    const int g_off = g_row * 128 + ((((grp & 1) * 16 + (li & 3) * 4) * 2 + wave_m * 64) ^ (((g_row >> 1) & 1) << 6));
    const int x_off0 = 64 * 128 + x_row * 128 + ((((grp & 1) * 16 + (li & 3) * 4) * 2 + wave_n * 64) ^ (((x_row >> 1) & 1) << 6));   // kh even
    const int x_off1 = x_off0 ^ 64;                                                                                                  // kh == 1
    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    typedef __attribute__((address_space(3))) wd_v4s* lp_t;
    auto tr4 = [&](const char* a) __attribute__((always_inline)) { return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)a)); };
    auto compute = [&](int stage, const ChunkAt& nx) __attribute__((always_inline)) {
        if (p.ablate & 2) {
            if (ILV) static_for(std::make_integer_sequence<int, 6>{}, [&](auto u_c) __attribute__((always_inline)) { issue_piece(nx, u_c); });
            return;
        }
        const char* sb = sm + stage * W9D_STAGE;
        static_for(std::make_integer_sequence<int, 4>{}, [&](auto ks_c) __attribute__((always_inline)) {
            constexpr int ks = decltype(ks_c)::value;
            const uint2 al = tr4(sb + g_off + ks * 16 * 128), ah = tr4(sb + g_off + ks * 16 * 128 + 4 * 128);
            const uint4 a = make_uint4(al.x, al.y, ah.x, ah.y);
            static_for(std::make_integer_sequence<int, 3>{}, [&](auto kh_c) __attribute__((always_inline)) {
                constexpr int kh = decltype(kh_c)::value, grp_i = ks * 3 + kh;      // MFMA group 0..11 of the chunk
                const char* rb = sb + (kh == 1 ? x_off1 : x_off0) + ((2 * ks + kh) * 10) * 128;
                const uint2 q0 = tr4(rb), q1 = tr4(rb + 4 * 128), q2 = tr4(rb + 8 * 128);
                if (ILV && grp_i % 2 == 0) issue_piece(nx, std::integral_constant<int, (grp_i / 2) % 6>{});    // every other group: no branch in the chunk
                const uint4 b0 = make_uint4(q0.x, q0.y, q1.x, q1.y);
                const uint4 b2 = make_uint4(q0.y, q1.x, q1.y, q2.x);
                const uint4 b1 = make_uint4(__builtin_amdgcn_alignbyte(q0.y, q0.x, 2), __builtin_amdgcn_alignbyte(q1.x, q0.y, 2),
                                            __builtin_amdgcn_alignbyte(q1.y, q1.x, 2), __builtin_amdgcn_alignbyte(q2.x, q1.y, 2));
                MmaT<DAT_BF16>::step(a, b0, acc[kh * 3 + 0]);
                MmaT<DAT_BF16>::step(a, b1, acc[kh * 3 + 1]);
                MmaT<DAT_BF16>::step(a, b2, acc[kh * 3 + 2]);
            });
        });
    };
    if (n_iter > 0) {
        issue(c_lo, 0);
        issue(c_lo + 1, 1);
        int stage = 0;
        for (unsigned it = 0; it < n_iter; ++it) {
            const unsigned c = c_lo + it;
            // all but the six pieces of chunk c + 1 have landed (mine of chunk c), my LDS reads of chunk c - 1 are done; then the barrier:
            // chunk c is complete for every wave and stage (c - 1) % 3 is free.  A bare s_barrier: __syncthreads() carries a fence that
            // the compiler lowers to vmcnt(0) -- it would wait for chunk c + 1 as well and serialise the pipeline again.
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int nstage = stage == 0 ? 2 : stage - 1;      // (c + 2) % 3 == (c - 1) % 3
            const ChunkAt nx = chunk_at(c + 2, nstage);
            if (!ILV) static_for(std::make_integer_sequence<int, 6>{}, [&](auto u_c) __attribute__((always_inline)) { issue_piece(nx, u_c); });
            compute(stage, nx);
            stage = stage == 2 ? 0 : stage + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the two trailing zero chunks: nothing may still be writing LDS at exit
    }
    const int khalf = lane >> 5;
    const int ci = ci_t * 64 + wave_n * 32 + (lane & 31);
    if (SUB == 2) {
        // ---- the two sub-blocks add their tiles through LDS: sub s keeps accumulator rows r = 8 s .. 8 s + 7 of every tap, hands the other eight
        //      to its partner (wave w of sub 1 - s holds the same 32 x 32 quadrant) -- 72 floats per lane each way, the 147 KB of the stages ----
        __syncthreads();                                        // every wave is done with the stages
        float4* mine = (float4*)(smem + (size_t)(sub * 4 + wave) * (18 * 64 * 16));
        const float4* theirs = (const float4*)(smem + (size_t)((sub ^ 1) * 4 + wave) * (18 * 64 * 16));
        auto put = [&](auto own_c) __attribute__((always_inline)) {
            constexpr int OTH = 1 - decltype(own_c)::value;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    mine[(t * 2 + q) * 64 + lane] = make_float4(acc[t][OTH * 8 + q * 4 + 0], acc[t][OTH * 8 + q * 4 + 1], acc[t][OTH * 8 + q * 4 + 2], acc[t][OTH * 8 + q * 4 + 3]);
        };
        auto get = [&](auto own_c) __attribute__((always_inline)) {
            constexpr int OWN = decltype(own_c)::value;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float* Gt = p.G + (size_t)(kt * 9 + t) * p.Cout * p.Cin;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 o = theirs[(t * 2 + q) * 64 + lane];
                    const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = OWN * 8 + q * 4 + e;
                        const int co = co_t * 64 + wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                        if (co >= p.Cout) continue;
                        const float v = acc[t][r] + ov[e];
                        if (p.atomic) atomicAdd(Gt + (size_t)co * p.Cin + ci, v);
                        else Gt[(size_t)co * p.Cin + ci] = v;
                    }
                }
            }
        };
        if (sub == 0) put(std::integral_constant<int, 0>{});
        else put(std::integral_constant<int, 1>{});
        __syncthreads();
        if (ci >= p.Cin || (p.ablate & 4)) return;
        if (sub == 0) get(std::integral_constant<int, 0>{});
        else get(std::integral_constant<int, 1>{});
        return;
    }
    if (ci >= p.Cin || (p.ablate & 4)) return;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* Gt = p.G + (size_t)(kt * 9 + t) * p.Cout * p.Cin;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_t * 64 + wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (co >= p.Cout) continue;
            if (p.atomic) atomicAdd(Gt + (size_t)co * p.Cin + ci, acc[t][r]);
            else Gt[(size_t)co * p.Cin + ci] = acc[t][r];
        }
    }
}

// ---- weight gradient of POINTWISE convs (1 x 1 x 1, FC; stride 1 | 2) on the MFMA pipe (round 5) ---------------------------------------
// wgrad_direct_kernel ran these layers (R-50's bottleneck 1 x 1 x 1 convs and shortcuts, the FPN laterals, the RPN / box heads) at an MFMA duty
// cycle of 0.11: a (128 co x 128 ci) tile reads 512 bytes per staged position for 32 KFLOP -- 64 FLOP per byte, every operand row
// fetched once per tile of the other operand (P x (Cout x n_ci + Cin x n_co) x 2 B: 4 x the compulsory traffic of a 128 -> 512 layer) --
// through a register-staged double buffer with ONE 32-KB chunk in flight per block.  Here:
//   * one EIGHT-wave block per CU owns a tile of 64 K accumulators -- 128 x 512, 256 x 256 or 512 x 128 (co x ci), whichever moves the
//     fewest bytes for the layer's channel counts; a wave owns 128 co x 64 ci (8 accumulator tiles, 128 registers): 12 transposing LDS
//     reads feed 8 MFMAs per 16-k step (the 128 x 128 kernel: 8 reads for 4);
//   * operands come in by LDS-DMA (`global_load_lds_dwordx4`, no staging registers) as 64-channel PANELS of 32 positions x 128 B into
//     THREE stages: chunk c computes while c + 1 and c + 2 are in flight (80 KB per CU), one counted `s_waitcnt vmcnt` + bare `s_barrier`
//     per chunk, the pieces of chunk c + 2 issued between the MFMA groups of chunk c -- the scheme of wgrad_dma9_kernel, whose swizzle
//     (16-byte piece p of row r in slot p ^ 4 * bit1(r)) keeps the transposing reads of four consecutive 128-byte rows on all 64 banks;
//   * K (positions) is split over ~one block per CU; partial tiles are added into Gt[co][ci] with float atomics.
struct WgradPwItem {
    const char* g;        // [frames * Ho * Wo][g_cs] bf16
    const char* x;        // [frames * H * W][x_cs] bf16
    float* G;             // [Cout][Cin] fp32
    int Cout, Cin, g_cs, x_cs;
    int H, W, Wo, stride;
    int wm;               // waves over the co axis: 1 | 2 | 4 -> tile = (128 wm) co x (64 * 8 / wm) ci
    int ksplit, n_ci_tiles, n_co_tiles;
    int atomic;
    unsigned p_begin, p_end;            // output positions [p_begin, p_end) carry a non-zero gradient
    unsigned how, how_magic, wo_magic;  // Ho * Wo; floor(2^32 / how) (quotient corrected by one step); ceil(2^32 / Wo)
    unsigned block0;                    // first block of the grid that works on this layer
};
// SEVERAL layers per launch (round 5).  Float atomics retire at about one dword per clock and L2 channel on this part (~1.2 TB/s: measured by
// sweeping the K split, tools/probes/wgrad_bench.py pw), and a layer split over all 256 CUs adds 256 partial tiles of 256 KB = 64 MB -- 50 us
// of atomics behind 20-30 us of MFMA work, the same for the 128 x 128 kernel this one replaces.  The partial volume is (blocks in flight) x
// (accumulators per block) whatever the tile, so the only way to shrink it is to give a layer FEWER blocks -- and the other CUs to other
// layers: the pointwise weight gradients of a whole gradient bucket (10-40 layers) are one grid of ~2 blocks per CU, every block a
// (layer, tile, K range) with about the same number of chunks.  The table travels in the kernel arguments (no device table to keep alive).
constexpr int PW_MAX_ITEMS = 24;
struct WgradPwBatch {
    const char* zeros;                  // >= 16 zero bytes in global memory
    int n;
    WgradPwItem item[PW_MAX_ITEMS];
};

constexpr int PW_KC = 32;                     // positions per chunk
constexpr int PW_PANEL = PW_KC * 128;         // one 64-channel panel of a chunk: 4 KB = four 1-KB DMA pieces

template <int PANELS>
__global__ __launch_bounds__(512, 1) void wgrad_pw_kernel(const WgradPwBatch b) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int STAGE = PANELS * PW_PANEL;
    constexpr int PPW = PANELS / 2;             // DMA pieces per wave and chunk (PANELS * 4 pieces over 8 waves)
    static_assert(PANELS == 8 || PANELS == 10, "tile shapes: 256 x 256 (4 + 4 panels), 128 x 512 / 512 x 128 (2 + 8 / 8 + 2)");
    int job = 0;
    while (job + 1 < b.n && blockIdx.x >= b.item[job + 1].block0) ++job;
    const WgradPwItem& p = b.item[job];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = p.wm;
    const int wm_i = wave & (wm - 1), wn_i = wave / wm;
    const int ncp = 2 * wm;                     // g panels of a stage; the x panels follow
    const int TM = 128 * wm, TN = 64 * (8 / wm);
    unsigned bid = blockIdx.x - p.block0;
    const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
    const int co_t = bid % p.n_co_tiles;
    const int split = bid / p.n_co_tiles;
    const unsigned p_begin = p.p_begin, p_end = p.p_end;
    const unsigned nchunks = (p_end - p_begin + (PW_KC - 1)) / PW_KC;
    const unsigned c_lo = (unsigned)((unsigned long long)nchunks * split / p.ksplit), c_hi = (unsigned)((unsigned long long)nchunks * (split + 1) / p.ksplit);
    const unsigned n_iter = c_hi - c_lo;
    const char* const gbase = p.g;
    const char* const xbase = p.x;
    const size_t g_pitch = (size_t)p.g_cs * 2, x_pitch = (size_t)p.x_cs * 2;
    const unsigned how = p.how, how_magic = p.how_magic, wo_magic = p.wo_magic, Wo = (unsigned)p.Wo, Hin = (unsigned)p.H, Win = (unsigned)p.W,
                   strd = (unsigned)p.stride;

    // ---- this lane's DMA pieces: piece gp = wave + 8 u of a stage covers rows (gp & 3) * 8 .. + 7 of panel gp >> 2; the lane writes
    //      row (gp & 3) * 8 + lane / 8, slot lane % 8, which holds data piece slot ^ 4 * bit1(row) ----
    int pk_row[PPW], pk_col[PPW];               // pk_col: byte offset of the piece inside the tensor's channel run, -1 = always zeros
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        const int gp = wave + 8 * u, panel = gp >> 2;
        const int row = (gp & 3) * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ (((row >> 1) & 1) << 2);
        int col = -1;
        if (panel < ncp) {
            const int ch = co_t * TM + panel * 64 + piece * 8;
            if (ch < p.g_cs) col = ch * 2;
        } else {
            const int ch = ci_t * TN + (panel - ncp) * 64 + piece * 8;
            if (ch < p.x_cs) col = ch * 2;
        }
        pk_row[u] = row; pk_col[u] = col;
    }
    const char* const zeros = b.zeros;
    auto issue_piece = [&](unsigned ch, int stage, auto u_c) __attribute__((always_inline)) {
        constexpr int u = decltype(u_c)::value;
        const int gp = wave + 8 * u;
        const bool is_g = (gp >> 2) < ncp;                              // (uniform per wave and u)
        const unsigned pos = p_begin + ch * (unsigned)PW_KC + (unsigned)pk_row[u];
        const bool ok = ch < c_hi && pos < p_end && pk_col[u] >= 0;
        const unsigned pc = min(pos, p_end - 1u);                       // (a clamped row: the address stays inside the tensor, the mask decides)
        // output position -> (frame, oy, ox) -> input row (frame, s oy, s ox); stride 1 maps a position onto itself (H x W = Ho x Wo)
        unsigned f = __umulhi(pc, how_magic);
        unsigned rem = pc - f * how;
        if (rem >= how) { rem -= how; ++f; }                            // (floor(2^32 / how) under-estimates the quotient by at most one)
        const unsigned oy = __umulhi(rem, wo_magic), ox = rem - oy * Wo;
        const size_t xrow = ((size_t)f * Hin + oy * strd) * Win + ox * strd;
        const char* at = (is_g ? gbase + (size_t)pc * g_pitch : xbase + xrow * x_pitch) + max(pk_col[u], 0);
        const unsigned long long m = ok ? ~0ull : 0ull;
        const char* src = (const char*)((unsigned long long)zeros + (((unsigned long long)at - (unsigned long long)zeros) & m));
        const unsigned lds_at = (unsigned)(size_t)(lptr_t)(smem + stage * STAGE + gp * 1024);
        // (inline asm without a memory clobber, M0 written behind the compiler's back: see wgrad_dma9_kernel)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_at));
    };
    auto issue = [&](unsigned ch, int stage) __attribute__((always_inline)) {
        static_for(std::make_integer_sequence<int, PPW>{}, [&](auto u_c) __attribute__((always_inline)) { issue_piece(ch, stage, u_c); });
        asm volatile("" ::: "memory");
    };
    // ---- fragment addressing: a 16-lane group reads the 4 (k) x 16 (channel) block at rows k0 + (grp >> 1) * 8 [+ 4], channels (grp & 1) * 16;
    //      lane i of the group supplies row i >> 2, channels 4 (i & 3) and receives channel i's four rows ----
    const int grp = lane >> 4, li = lane & 15;
    const int f_row = (grp >> 1) * 8 + (li >> 2);
    const int f_col = (((grp & 1) * 16 + (li & 3) * 4) * 2) ^ (((f_row >> 1) & 1) << 6);
    int a_off[4], b_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[i] = (wm_i * 2 + (i >> 1)) * PW_PANEL + f_row * 128 + (f_col ^ ((i & 1) << 6));
#pragma unroll
    for (int j = 0; j < 2; ++j) b_off[j] = (ncp + wn_i) * PW_PANEL + f_row * 128 + (f_col ^ (j << 6));
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    typedef __attribute__((address_space(3))) wd_v4s* lp_t;
    auto frag = [&](const char* a) __attribute__((always_inline)) {
        const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)a));
        const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(a + 4 * 128)));
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    auto compute = [&](int stage, unsigned nx_ch, int nx_stage) __attribute__((always_inline)) {
        const char* sb = smem + stage * STAGE;
        uint4 a[2][4], bb[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[0][i] = frag(sb + a_off[i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bb[0][j] = frag(sb + b_off[j]);
        static_for(std::make_integer_sequence<int, PW_KC / 16>{}, [&](auto ks_c) __attribute__((always_inline)) {
            constexpr int ks = decltype(ks_c)::value;
            if (ks + 1 < PW_KC / 16) {          // the fragments of k-step ks + 1 are requested before the MFMAs of step ks
#pragma unroll
                for (int i = 0; i < 4; ++i) a[(ks + 1) & 1][i] = frag(sb + a_off[i] + (ks + 1) * 16 * 128);
#pragma unroll
                for (int j = 0; j < 2; ++j) bb[(ks + 1) & 1][j] = frag(sb + b_off[j] + (ks + 1) * 16 * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for(std::make_integer_sequence<int, 4>{}, [&](auto i_c) __attribute__((always_inline)) {
                constexpr int i = decltype(i_c)::value, slot = ks * 4 + i;     // MFMA pair 0..7 of the chunk
                if (slot < PPW) issue_piece(nx_ch, nx_stage, std::integral_constant<int, slot < PPW ? slot : 0>{});
                MmaT<DAT_BF16>::step(a[ks & 1][i], bb[ks & 1][0], acc[i][0]);
                MmaT<DAT_BF16>::step(a[ks & 1][i], bb[ks & 1][1], acc[i][1]);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    if (n_iter > 0) {
        issue(c_lo, 0);
        issue(c_lo + 1, 1);
        int stage = 0;
        for (unsigned it = 0; it < n_iter; ++it) {
            // all but my PPW pieces of chunk c + 1 have landed and my LDS reads of chunk c - 1 are done; behind the barrier chunk c is complete
            // for every wave and stage (c - 1) % 3 is free for chunk c + 2 (a bare s_barrier: __syncthreads() would wait for vmcnt(0))
            if (PPW == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int nstage = stage == 0 ? 2 : stage - 1;
            compute(stage, c_lo + it + 2, nstage);
            stage = stage == 2 ? 0 : stage + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the two trailing (zero) chunks: nothing may still be writing LDS at exit
    }
    const int khalf = lane >> 5;
    const int Cout = p.Cout, Cin = p.Cin, atomic = p.atomic;
    float* const G = p.G;
    if (n_iter == 0 && atomic) return;          // (nothing to add)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci_t * TN + wn_i * 64 + j * 32 + (lane & 31);
            if (ci >= Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_t * TM + wm_i * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (co >= Cout) continue;
                if (atomic) atomicAdd(G + (size_t)co * Cin + ci, acc[i][j][r]);
                else G[(size_t)co * Cin + ci] = acc[i][j][r];
            }
        }
}

// Gt[tap][co][ci] -> dW[co][ci][tap] * scale[co] (the fused AffineChannelNd scale; NULL = 1)
__global__ void wgrad_finish_kernel(const float* __restrict__ Gt, const float* __restrict__ scale, float* __restrict__ dW, int Cout,
                                    int Cin, int ntaps) {
    const long long total = (long long)Cout * Cin * ntaps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % ntaps);
        const long long r = i / ntaps;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        const float v = Gt[((size_t)tap * Cout + co) * Cin + ci];
        dW[i] = scale ? v * scale[co] : v;
    }
}

// ---- elementwise backward pieces -----------------------------------------------------------------------------------------------
// g = dy * (y > 0) (Relu backward on the fused conv's OUTPUT, exact because y == 0 exactly where the ReLU clipped),
// optionally + dy2 first (a blob read by two consumers); channel sums of g accumulate into dbias (AffineChannelNd / conv
// bias gradient, affine_channel_nd_op.cu:74-92 computes the same reduction for its bias).
template <int DT>
__global__ void relu_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ dy2, const void* __restrict__ y,
                                void* __restrict__ g, float* __restrict__ dbias, float* __restrict__ partial, long long npos, int C, int cs,
                                int relu) {
    // thread = one channel quad of a strip of positions; blockDim.x = cs/4 quads... generic: grid-stride over (pos, c4)
    const int nq = cs / 4;
    const long long total = npos * nq;
    __shared__ float red[1024 * 4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const long long stride = (long long)gridDim.x * blockDim.x;
    // make every thread own ONE channel quad so that its partial sums are per-channel: requires stride % nq == 0
    // stride % nq == 0: the thread's channel quad is fixed, only the position advances (no division in the loop)
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(gid % nq) * 4;
    const long long pstep = stride / nq;
    (void)total;
    // U positions per trip, all loads of a trip issued before their first use: with <= 512 blocks (bias-reducing launches) a thread
    // has 20-40 trips, and one 8-byte load in flight per trip made the launch a chain of memory round trips (77 us for a 20 MB
    // tensor = 0.26 TB/s, profiles/r03/train_r18 kernel trace)
    constexpr int U = 4;
    for (long long pos0 = gid / nq; pos0 < npos; pos0 += pstep * U) {
        float vv[U][4], oo[U][4];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pos = pos0 + pstep * u;
            live[u] = pos < npos;
            const size_t at = (size_t)(live[u] ? pos : pos0) * cs + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) oo[u][e] = 1.f;
            if (DT == DAT_BF16) {
                const uint2 a = ((const uint2*)dy)[at >> 2];
                vv[u][0] = bf2f((uint16_t)(a.x & 0xffff)); vv[u][1] = bf2f((uint16_t)(a.x >> 16));
                vv[u][2] = bf2f((uint16_t)(a.y & 0xffff)); vv[u][3] = bf2f((uint16_t)(a.y >> 16));
                if (dy2) {
                    const uint2 b = ((const uint2*)dy2)[at >> 2];
                    vv[u][0] += bf2f((uint16_t)(b.x & 0xffff)); vv[u][1] += bf2f((uint16_t)(b.x >> 16));
                    vv[u][2] += bf2f((uint16_t)(b.y & 0xffff)); vv[u][3] += bf2f((uint16_t)(b.y >> 16));
                }
                if (relu) {
                    const uint2 q = ((const uint2*)y)[at >> 2];
                    oo[u][0] = bf2f((uint16_t)(q.x & 0xffff)); oo[u][1] = bf2f((uint16_t)(q.x >> 16));
                    oo[u][2] = bf2f((uint16_t)(q.y & 0xffff)); oo[u][3] = bf2f((uint16_t)(q.y >> 16));
                }
            } else {
                const float4 a = ((const float4*)dy)[at >> 2];
                vv[u][0] = a.x; vv[u][1] = a.y; vv[u][2] = a.z; vv[u][3] = a.w;
                if (dy2) {
                    const float4 b = ((const float4*)dy2)[at >> 2];
                    vv[u][0] += b.x; vv[u][1] += b.y; vv[u][2] += b.z; vv[u][3] += b.w;
                }
                if (relu) {
                    const float4 q = ((const float4*)y)[at >> 2];
                    oo[u][0] = q.x; oo[u][1] = q.y; oo[u][2] = q.z; oo[u][3] = q.w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            const size_t at = (size_t)(pos0 + pstep * u) * cs + c;
            float* v = vv[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if ((relu && !(oo[u][e] > 0.f)) || c + e >= C) v[e] = 0.f;
                s[e] += v[e];
            }
            if (!g) continue;      // reduction only (no ReLU, no padded channels: the caller keeps using dy itself)
            if (DT == DAT_BF16) {
                uint2 w;
                w.x = f2bf2(v[0], v[1]);
                w.y = f2bf2(v[2], v[3]);
                ((uint2*)g)[at >> 2] = w;
            } else {
                ((float4*)g)[at >> 2] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    if (!dbias) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = s[e];
    __syncthreads();
    // threads with the same (threadIdx.x % nq) hold the same channel quad (blockDim.x % nq == 0 enforced by the host)
    if ((int)threadIdx.x < nq) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = threadIdx.x; j < (int)blockDim.x; j += nq)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += red[j * 4 + e];
        const int c = threadIdx.x * 4;
        // per-block partial sums, reduced by bias_partial_reduce_kernel: one float atomic per channel per block on the few cache lines
        // of dbias serialised in L2 (512 blocks x 512 channels: ~75 us of a 77-us launch over a 25 MB tensor)
        *(float4*)(partial + (size_t)blockIdx.x * cs + c) = make_float4(t[0], t[1], t[2], t[3]);
    }
}

// dbias[c] += sum_b partial[b][c]: a block = 32 consecutive channels x 32 row lanes (rows rl, rl + 32, ...; eight independent loads in flight
// per thread), then an LDS reduction over the row lanes.  (Round 4: 32 row lanes instead of 8 -- with up to 2048 partial rows a thread of the
// 8-lane version walked 32 dependent rounds of loads, ~10 us per launch, 31 launches per training iteration.)
constexpr int BPR_LANES = 32;
__global__ __launch_bounds__(32 * BPR_LANES) void bias_partial_reduce_kernel(const float* __restrict__ partial, int nblocks, int cs, int C,
                                                                           float* __restrict__ dbias) {
    __shared__ float red[BPR_LANES][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < cs) {
        int b = rl;
        for (; b + 7 * BPR_LANES < nblocks; b += 8 * BPR_LANES) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += partial[(size_t)(b + BPR_LANES * u) * cs + c];
        }
        for (; b < nblocks; b += BPR_LANES) acc[0] += partial[(size_t)b * cs + c];
    }
    red[rl][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (rl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < BPR_LANES; ++r) t += red[r][cl];
        dbias[c] += t;
    }
}

// dst[f, 2y, 2x, :] = src[f, y, x, :], zeros elsewhere (input of the stride-2 data gradient run as a stride-1 conv)
template <int DT>
__global__ void zero_insert2x_kernel(const void* __restrict__ src, void* __restrict__ dst, int frames, int Hs, int Ws, int Hd,
                                     int Wd, int cs) {
    const long long total = (long long)frames * Hd * Wd * (cs / 4);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (cs / 4)) * 4;
        long long r = i / (cs / 4);
        const int x = (int)(r % Wd); r /= Wd;
        const int y = (int)(r % Hd);
        const int f = (int)(r / Hd);
        const bool live = !(x & 1) && !(y & 1) && (y >> 1) < Hs && (x >> 1) < Ws;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = 0.f;
            if (live) v = ElemOf<DT>::ld(src, (((size_t)f * Hs + (y >> 1)) * Ws + (x >> 1)) * cs + c + e);
            ElemOf<DT>::st(dst, (((size_t)f * Hd + y) * Wd + x) * cs + c + e, v);
        }
    }
}

// FPN top-down backward (FPN3D.py:207-222: UpsampleNearest x2 + Sum): dtop[f, y, x, :] (+)= sum of the 2x2 block of g
template <int DT>
__global__ void upsample2x_bwd_kernel(const void* __restrict__ g, void* __restrict__ dtop, int frames, int Ht, int Wt, int cs,
                                      int accumulate) {
    const long long total = (long long)frames * Ht * Wt * cs;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cs);
        long long r = i / cs;
        const int x = (int)(r % Wt); r /= Wt;
        const int y = (int)(r % Ht);
        const int f = (int)(r / Ht);
        const size_t W2 = 2 * (size_t)Wt;
        const size_t b = (((size_t)f * 2 * Ht + 2 * y) * W2 + 2 * x) * cs + c;
        float v = ElemOf<DT>::ld(g, b) + ElemOf<DT>::ld(g, b + cs) + ElemOf<DT>::ld(g, b + W2 * cs) + ElemOf<DT>::ld(g, b + W2 * cs + cs);
        if (accumulate) v += ElemOf<DT>::ld(dtop, i);
        ElemOf<DT>::st(dtop, i, v);
    }
}

// MomentumSGDUpdate with the reference's pre-processing (model_builder.py:954-985): biases: grad *= 2, no decay;
// weights: grad += wd * w;  v = mu*v + lr*grad;  w -= v.
__device__ __forceinline__ void sgd_one(float& w, float& v, float g, float lr, float mu, float wd, int is_bias) {
    g = is_bias ? 2.f * g : g + wd * w;
    const float nv = mu * v + lr * g;
    v = nv;
    w -= nv;
}
// 16-byte accesses over the part of the three arrays that is 16-byte aligned in all of them (the flat arenas are sliced at the same element
// offsets, so they share their misalignment): the scalar form moved its 20 bytes per parameter at 2.6 TB/s.  Per-element arithmetic unchanged.
__global__ void sgd_momentum_kernel(float* __restrict__ w, float* __restrict__ v, const float* __restrict__ grad, long long n,
                                    float lr, float mu, float wd, int is_bias) {
    const unsigned long long aw = (unsigned long long)w, av = (unsigned long long)v, ag = (unsigned long long)grad;
    const bool same = ((aw ^ av) & 15) == 0 && ((aw ^ ag) & 15) == 0;
    long long head = same ? (long long)(((16 - (aw & 15)) >> 2) & 3) : n;
    if (head > n) head = n;
    const long long n4 = (n - head) / 4;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    float4* w4 = (float4*)(w + head);
    float4* v4 = (float4*)(v + head);
    const float4* g4 = (const float4*)(grad + head);
    for (long long i = gid; i < n4; i += stride) {
        float4 ww = w4[i], vv = v4[i];
        const float4 gg = g4[i];
        sgd_one(ww.x, vv.x, gg.x, lr, mu, wd, is_bias);
        sgd_one(ww.y, vv.y, gg.y, lr, mu, wd, is_bias);
        sgd_one(ww.z, vv.z, gg.z, lr, mu, wd, is_bias);
        sgd_one(ww.w, vv.w, gg.w, lr, mu, wd, is_bias);
        v4[i] = vv;
        w4[i] = ww;
    }
    // the unaligned head and the tail (at most 3 + 3 elements; everything when the arrays are misaligned differently)
    const long long tail0 = head + n4 * 4;
    for (long long i = gid; i < head + (n - tail0); i += stride) {
        const long long j = i < head ? i : tail0 + (i - head);
        sgd_one(w[j], v[j], grad[j], lr, mu, wd, is_bias);
    }
}

// ---- RoIAlign backward: scatter the cell gradient over the bilinear taps of its samples (fp32 atomics) -------------------------
// Mirrors roi_align.hip (legacy Detectron RoIAlign, FPN level picked per RoI in-kernel) sample for sample.
struct RoiBwdParams {
    float* dfeat[4];      // fp32 [frames, H, W, C] per level, accumulated into
    int H[4], W[4];
    float scale[4];
    int n_levels, k_min, canon_level;
    float canon_scale;
    int T, C;
    const float* rois;
    int R, Tr, t0, pooled, sampling;
    const char* dout;     // [R*Tr, P, P, C] in DT
    int fold;             // fold the samples of a bin into per-pixel weights before the atomics
};

template <int DT>
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const RoiBwdParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = p.pooled;
    const long long ncell = (long long)p.R * p.Tr * P * P;
    const int roi_cols = 4 * p.Tr + 1;
    for (long long cell = (long long)blockIdx.x * 4 + wave; cell < ncell; cell += (long long)gridDim.x * 4) {
        const int pw = cell % P;
        long long q = cell / P;
        const int ph = q % P; q /= P;
        const int t = q % p.Tr;
        const int r = q / p.Tr;
        const float* roi = p.rois + (size_t)r * roi_cols;
        const int n = (int)roi[0];
        const float bx1 = roi[1 + 4 * t], by1 = roi[2 + 4 * t], bx2 = roi[3 + 4 * t], by2 = roi[4 + 4 * t];
        int lvl = 0;
        if (p.n_levels > 1) {
            float asum = 0.f;
            for (int tt = 0; tt < p.Tr; ++tt) {
                const float w = roi[3 + 4 * tt] - roi[1 + 4 * tt] + 1.f;
                const float h = roi[4 + 4 * tt] - roi[2 + 4 * tt] + 1.f;
                asum += w * h;
            }
            const float sarea = sqrtf(asum / (float)p.Tr);
            float l = floorf((float)p.canon_level + log2f(sarea / p.canon_scale + 1e-6f));
            l = fminf(fmaxf(l, (float)p.k_min), (float)(p.k_min + p.n_levels - 1));
            lvl = (int)l - p.k_min;
        }
        const int H = p.H[lvl], W = p.W[lvl];
        const float sc = p.scale[lvl];
        const int frame = n * p.T + (p.Tr == 1 ? p.t0 : t);
        float* fbase = p.dfeat[lvl] + (size_t)frame * H * W * p.C;
        const float x1 = bx1 * sc, y1 = by1 * sc, x2 = bx2 * sc, y2 = by2 * sc;
        const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
        const float bw = rw / (float)P, bh = rh / (float)P;
        const int gh = p.sampling > 0 ? p.sampling : (int)ceilf(rh / (float)P);
        const int gw = p.sampling > 0 ? p.sampling : (int)ceilf(rw / (float)P);
        const float inv = 1.f / (float)(gh * gw);
        const size_t obase = ((((size_t)r * p.Tr + t) * P + ph) * P + pw) * p.C;
        // The bilinear weights of a bin do not depend on the channel: the gh x gw samples of the bin (2 x 2 at the reference's sampling
        // ratio) are folded into a weight per distinct pixel BEFORE the channel loop -- they cover at most (gh + 1) x (gw + 1) pixels,
        // often 2 x 2 -- and every lane then issues one atomic per pixel and channel instead of four per SAMPLE and channel (16 -> 4..9
        // for 2 x 2 samples: the float atomics of overlapping rois serialise on the same lines and were the whole 0.27 ms of a launch).
        // The fold is wave-uniform work: all 64 lanes execute it in lock step on the same values (one instruction stream per cell, not
        // one per channel); its two small tables are indexed at run time and therefore live in scratch (~30 scratch accesses
        // per cell, against 64 x (4..9) atomics saved -- a register-only de-duplication was not worth its code).
        constexpr int MAXPIX = 16;
        int pix[MAXPIX];
        float wgt[MAXPIX];
        int npix = 0;
        bool folded = p.fold && gh * gw <= 4;
        if (folded) {
            for (int iy = 0; iy < gh; ++iy) {
                const float y = y1 + (float)ph * bh + ((float)iy + .5f) * bh / (float)gh;
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
                    const int pidx[4] = {yl * W + xl, yl * W + xh, yh * W + xl, yh * W + xh};
                    const float pw4[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        int at = -1;
                        for (int u = 0; u < npix; ++u)
                            if (pix[u] == pidx[k]) at = u;
                        if (at < 0) { at = npix++; pix[at] = pidx[k]; wgt[at] = 0.f; }
                        wgt[at] += pw4[k];
                    }
                }
            }
        }
        for (int c = lane; c < p.C; c += 64) {
            const float g = ElemOf<DT>::ld(p.dout, obase + c) * inv;
            if (g == 0.f) continue;
            if (folded) {
                for (int u = 0; u < npix; ++u)
                    if (wgt[u] != 0.f) atomicAdd(fbase + (size_t)pix[u] * p.C + c, g * wgt[u]);
                continue;
            }
            for (int iy = 0; iy < gh; ++iy) {
                const float y = y1 + (float)ph * bh + ((float)iy + .5f) * bh / (float)gh;
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
                    atomicAdd(fbase + ((size_t)yl * W + xl) * p.C + c, g * hy * hx);
                    atomicAdd(fbase + ((size_t)yl * W + xh) * p.C + c, g * hy * lx);
                    atomicAdd(fbase + ((size_t)yh * W + xl) * p.C + c, g * ly * hx);
                    atomicAdd(fbase + ((size_t)yh * W + xh) * p.C + c, g * ly * lx);
                }
            }
        }
    }
}

// ---- backward of dat_kps_finalize: dsub[fr, y', x', (a*2+b)*K + k] = sum_{oy,ox} dout[r, t*K+k, oy, ox] f(oy) f(ox) ---------------
template <int DT>
__global__ void kps_finalize_bwd_kernel(const float* __restrict__ dout, int R, int Tr, int S, int cs, int K, int up,
                                        void* __restrict__ dsub) {
    const int L = 2 * S, M = L * up;
    const int ksz = 2 * up, pad = up / 2;
    const float factor = (float)((ksz + 1) / 2);
    const float center = (ksz % 2 == 1) ? factor - 1.f : factor - 0.5f;
    const size_t total = (size_t)R * Tr * S * S * cs;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = i % cs;
        size_t q = i / cs;
        const int xs = q % S; q /= S;
        const int ys = q % S;
        const size_t fr = q / S;
        float acc = 0.f;
        if (ch < 4 * K) {
            const int k = ch % K, ab = ch / K;
            const int y = 2 * ys + (ab >> 1), x = 2 * xs + (ab & 1);
            const size_t r = fr / Tr, t = fr % Tr;
            const float* src = dout + ((r * Tr + t) * K + k) * (size_t)M * M;
            for (int ky = 0; ky < ksz; ++ky) {
                const int oy = up * y - pad + ky;
                if (oy < 0 || oy >= M) continue;
                const float fy = 1.f - fabsf((float)ky - center) / factor;
                for (int kx = 0; kx < ksz; ++kx) {
                    const int ox = up * x - pad + kx;
                    if (ox < 0 || ox >= M) continue;
                    const float fx = 1.f - fabsf((float)kx - center) / factor;
                    acc += src[(size_t)oy * M + ox] * (fy * fx);
                }
            }
        }
        ElemOf<DT>::st(dsub, i, acc);
    }
}

static inline int grid_for(long long n, int block) {
    long long b = (n + block - 1) / block;
    return (int)(b > 65535 ? 65535 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

// geometry shared by the workspace query and the launch
static void wgrad_geom(const dat_conv_desc* d, int* Ho, int* Wo, int* Tq, int* Hq, int* Wq, long long* Qtot, long long* Qa,
                       int* ncopies) {
    dat_conv3d_out_shape(d, Ho, Wo);
    const int s = d->stride_h;
    *Tq = d->T + d->KT - 1;
    *Hq = *Ho + (d->KH - 1) / s;
    *Wq = *Wo + (d->KW - 1) / s;
    if (*Wq & 1) ++*Wq;
    const int clips = d->frames / d->T;
    *Qtot = (long long)clips * *Tq * *Hq * *Wq;
    const long long max_off = ((long long)(d->KT - 1) * *Hq + (d->KH - 1) / s) * *Wq + (d->KW - 1) / s;
    const int ck = d->dtype == DAT_BF16 ? 64 : 32;
    long long k = (*Qtot + ck - 1) / ck * ck;
    *Qa = k + (max_off + 63) / 64 * 64 + 64;
    *ncopies = (d->dtype == DAT_BF16 && (d->KW - 1) / s >= 1) ? 2 : 1;
}

size_t dat_conv3d_wgrad_workspace_bytes(const dat_conv_desc* d, int Cin_real, int Cout_real) {
    int Ho, Wo, Tq, Hq, Wq, nc;
    long long Qtot, Qa;
    wgrad_geom(d, &Ho, &Wo, &Tq, &Hq, &Wq, &Qtot, &Qa, &nc);
    const int s = d->stride_h;
    const size_t es = dat_esize(d->dtype);
    return ((size_t)Cout_real + (size_t)Cin_real * s * s * nc) * (size_t)Qa * es + 512 +
           (size_t)Cout_real * Cin_real * d->KT * d->KH * d->KW * sizeof(float);
}

static bool wgrad_direct_eligible(const dat_ctx* ctx, const dat_conv_desc* d, int g_cstride) {
    int Ho, Wo;
    dat_conv3d_out_shape(d, &Ho, &Wo);
    const long long npos_out = (long long)d->frames * Ho * Wo, npos_in = (long long)d->frames * d->H * d->W;
    return d->dtype == DAT_BF16 && ctx->dbg_wgrad_direct && d->Cin % 64 == 0 && g_cstride % 64 == 0 && npos_out < (1ll << 31) && npos_in < (1ll << 31) &&
           (long long)Ho * Wo * Wo < (1ll << 32);
}

// ---- pointwise layers: plan (tile shape, geometry) and grouped launch ------------------------------------------------------------
struct PwPlan {
    WgradPwItem it;
    long long tiles, nchunks;
};

static bool pw_eligible(const dat_ctx* ctx, const dat_conv_desc* d) {
    return d->KT == 1 && d->KH == 1 && d->KW == 1 && d->pad_h == 0 && d->pad_w == 0 && d->pad_t == 0 && ctx->dbg_wgrad_pw;
}

static int pw_plan(dat_ctx* ctx, const dat_conv_desc* d, const void* x, const void* g, int g_cstride, int Cin_real, int Cout_real, float* Gt,
                   PwPlan* pl) {
    int Ho, Wo;
    dat_conv3d_out_shape(d, &Ho, &Wo);
    const int clips = d->frames / d->T;
    WgradPwItem& q = pl->it;
    memset(&q, 0, sizeof(q));
    q.g = (const char*)g; q.x = (const char*)x; q.G = Gt;
    q.Cout = Cout_real; q.Cin = Cin_real; q.g_cs = g_cstride; q.x_cs = d->Cin;
    q.H = d->H; q.W = d->W; q.Wo = Wo; q.stride = d->stride_h;
    q.how = (unsigned)(Ho * Wo);
    q.how_magic = q.how == 1 ? 0xffffffffu : (unsigned)(0x100000000ull / q.how);
    q.wo_magic = Wo == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)Wo - 1) / (unsigned)Wo);
    q.p_begin = 0; q.p_end = (unsigned)((long long)d->frames * Ho * Wo);
    if (d->out_tn > 0 && clips == 1) {      // g is non-zero only in frames [out_t0, out_t0 + out_tn) of the clip
        DAT_ENFORCE(ctx, d->out_t0 >= 0 && d->out_t0 + d->out_tn <= d->T, "conv3d_wgrad: gradient frames [%d, %d) outside T %d",
                    d->out_t0, d->out_t0 + d->out_tn, d->T);
        q.p_begin = (unsigned)d->out_t0 * q.how; q.p_end = (unsigned)(d->out_t0 + d->out_tn) * q.how;
    }
    // tile shape: least padded MFMA work first, then least operand traffic (every g row is read once per ci tile, every x row once per
    // co tile); ties go to the squarer tile
    int best_wm = 2;
    long long best_work = -1, best_traffic = 0;
    const int wm_order[3] = {2, 1, 4};
    for (int k = 0; k < 3; ++k) {
        const int wm = ctx->dbg_wgrad_pw >= 10 ? ctx->dbg_wgrad_pw / 10 : wm_order[k];      // (DAT_WGRAD_PW = 10 / 20 / 40: forced shape)
        const long long tm = 128 * wm, tn = 64 * (8 / wm);
        const long long nco = (Cout_real + tm - 1) / tm, nci = (Cin_real + tn - 1) / tn;
        const long long work = nco * tm * nci * tn, traffic = (long long)g_cstride * nci + (long long)d->Cin * nco;
        if (best_work < 0 || work < best_work || (work == best_work && traffic < best_traffic)) {
            best_work = work; best_traffic = traffic; best_wm = wm;
        }
    }
    q.wm = best_wm;
    const int tm = 128 * q.wm, tn = 64 * (8 / q.wm);
    q.n_co_tiles = (Cout_real + tm - 1) / tm; q.n_ci_tiles = (Cin_real + tn - 1) / tn;
    pl->tiles = (long long)q.n_co_tiles * q.n_ci_tiles;
    pl->nchunks = ((long long)(q.p_end - q.p_begin) + PW_KC - 1) / PW_KC;
    return DAT_OK;
}

// launches the planned items (ksplit / atomic set by the caller): one grid per tile class (8 | 10 panels per stage), <= PW_MAX_ITEMS layers each
static int pw_launch(dat_ctx* ctx, hipStream_t st, PwPlan* pls, int n) {
    for (int cls = 0; cls < 2; ++cls) {
        int i = 0;
        while (i < n) {
            WgradPwBatch b;
            b.zeros = (const char*)ctx->zeros;
            b.n = 0;
            unsigned blocks = 0;
            for (; i < n && b.n < PW_MAX_ITEMS; ++i) {
                if ((pls[i].it.wm == 2) != (cls == 0)) continue;
                WgradPwItem& q = b.item[b.n++];
                q = pls[i].it;
                q.block0 = blocks;
                blocks += (unsigned)(pls[i].tiles * q.ksplit);
            }
            if (b.n == 0) continue;
            if (cls == 0) {
                const size_t lds = 3 * (size_t)8 * PW_PANEL;
                if (dat_ensure_lds(ctx, (const void*)wgrad_pw_kernel<8>, (int)lds) != DAT_OK) return DAT_ERR_LAUNCH;
                hipLaunchKernelGGL((wgrad_pw_kernel<8>), dim3(blocks), dim3(512), lds, st, b);
            } else {
                const size_t lds = 3 * (size_t)10 * PW_PANEL;
                if (dat_ensure_lds(ctx, (const void*)wgrad_pw_kernel<10>, (int)lds) != DAT_OK) return DAT_ERR_LAUNCH;
                hipLaunchKernelGGL((wgrad_pw_kernel<10>), dim3(blocks), dim3(512), lds, st, b);
            }
        }
    }
    DAT_CHECK_LAUNCH(ctx, "conv3d_wgrad pointwise batch");
    return DAT_OK;
}

// acc_mode (dat_conv3d_wgrad_acc): `workspace` IS the caller's fp32 accumulator in the kernels' [tap][Cout][Cin] order -- no zeroing, always
// atomics, no finish launch (the caller zeroes it once per iteration and runs ONE dat_wgrad_finish_batch for all layers)
static int wgrad_impl(dat_ctx* ctx, dat_stream s_, const dat_conv_desc* d, const void* x, const void* g, int g_cstride,
                      int Cin_real, int Cout_real, const float* scale, void* workspace, float* dW, int acc_mode) {
    DAT_ENFORCE(ctx, d && x && g && workspace && (dW || acc_mode), "conv3d_wgrad: null argument");
    DAT_ENFORCE(ctx, d->dtype == DAT_F32 || d->dtype == DAT_BF16, "conv3d_wgrad: bad dtype %d", d->dtype);
    DAT_ENFORCE(ctx, d->stride_h == d->stride_w && (d->stride_h == 1 || d->stride_h == 2), "conv3d_wgrad: stride %dx%d", d->stride_h, d->stride_w);
    DAT_ENFORCE(ctx, d->frames % d->T == 0 && d->pad_t * 2 + 1 == d->KT, "conv3d_wgrad: needs same-T convs");
    DAT_ENFORCE(ctx, d->KT * d->KH * d->KW <= 32, "conv3d_wgrad: %d taps exceed 32", d->KT * d->KH * d->KW);
    DAT_ENFORCE(ctx, d->Cin % 8 == 0 && g_cstride % 8 == 0, "conv3d_wgrad: channel strides must be multiples of 8");
    hipStream_t st = (hipStream_t)s_;
    int Ho, Wo, Tq, Hq, Wq, nc;
    long long Qtot, Qa;
    wgrad_geom(d, &Ho, &Wo, &Tq, &Hq, &Wq, &Qtot, &Qa, &nc);
    const int s = d->stride_h;
    const size_t es = dat_esize(d->dtype);
    const int clips = d->frames / d->T;
    // ---- bf16: the direct kernel (no re-pack passes; transposing LDS reads) ----
    const long long npos_out = (long long)d->frames * Ho * Wo, npos_in = (long long)d->frames * d->H * d->W;
    (void)npos_out; (void)npos_in;
    DAT_ENFORCE(ctx, !acc_mode || wgrad_direct_eligible(ctx, d, g_cstride), "conv3d_wgrad_acc: the layer does not take the direct kernels (ask dat_conv3d_wgrad_acc_supported)");
    if (wgrad_direct_eligible(ctx, d, g_cstride)) {
        float* Gt = (float*)workspace;
        if (d->KH == 3 && d->KW == 3 && s == 1 && d->pad_h == 1 && d->pad_w == 1 && (ctx->dbg_wgrad_direct & 2) == 0) {
            // nine spatial taps from one staged patch (wgrad_direct9_kernel)
            Wgrad9Params q;
            memset(&q, 0, sizeof(q));
            q.g = (const char*)g; q.x = (const char*)x; q.G = Gt;
            q.Cout = Cout_real; q.Cin = Cin_real; q.g_cs = g_cstride; q.x_cs = d->Cin;
            q.T = d->T; q.H = d->H; q.W = d->W; q.KT = d->KT; q.pt = d->pad_t;
            q.n_co_tiles = (Cout_real + 63) / 64; q.n_ci_tiles = (Cin_real + 63) / 64;
            q.f_begin = 0; q.f_end = d->frames;
            if (d->out_tn > 0 && clips == 1) {
                DAT_ENFORCE(ctx, d->out_t0 >= 0 && d->out_t0 + d->out_tn <= d->T, "conv3d_wgrad: gradient frames [%d, %d) outside T %d",
                            d->out_t0, d->out_t0 + d->out_tn, d->T);
                q.f_begin = d->out_t0; q.f_end = d->out_t0 + d->out_tn;
            }
            q.tiles_h = (d->H + 7) / 8; q.tiles_w = (d->W + 7) / 8;
            q.ablate = ctx->dbg_ablate_wgrad;
            const long long tiles = (long long)d->KT * q.n_co_tiles * q.n_ci_tiles;
            const long long nchunks = (long long)(q.f_end - q.f_begin) * q.tiles_h * q.tiles_w;
            // K split (per-layer sweep of the LDS-DMA kernel, tools/probes/wgrad_bench.py, blocks = tiles x ks): ~384 blocks -- 1.5 per CU --
            // beat the 640 (1.25 rounds of the 512 slots) the register-staged kernel liked: res3 (12 tiles) ks 53 -> 32 0.181 -> 0.157 ms,
            // res4 / P3 (48) 13 -> 8 0.192 -> 0.167 / 0.136 -> 0.109, P2 13 -> 8 0.328 -> 0.304; fewer, longer blocks mean fewer atomics and
            // fewer pipeline fills.  Layers with many tiles (res5: 192) want two blocks per CU again: ks 3 -> 4 0.224 -> 0.205.
            // Eight-wave blocks (DAT_WGRAD_SUB = 2, the default; round 4): one block per CU, two K ranges per block -- ks counts the RANGES, so it
            // is even; 2 * floor(256 / tiles) was the best of a sweep for every layer with <= 48 tiles (res3 0.164 -> 0.145 ms, res4 0.166 ->
            // 0.157, P2 0.306 -> 0.274, P3 0.109 -> 0.104).  Layers with more tiles than half the CUs (res5: 192) take ONE block per tile with
            // its two ranges -- no K split across blocks, so no contended atomics at all (plain stores outside the deferred-finish mode):
            // 0.210 -> 0.192 ms against the four-wave blocks' ks = 4 (768 blocks, 113 MB of atomics); two blocks per tile: 0.230.
            const int sub = ctx->dbg_wgrad_dma && ctx->dbg_wgrad_sub >= 2 ? 2 : 1;
            long long ks = ctx->dbg_wgrad_dma ? (tiles <= 96 ? (384 + tiles - 1) / tiles : (768 + tiles - 1) / tiles) : 640 / tiles;
            if (sub == 2) ks = tiles <= 128 ? 2 * (256 / tiles) : 2;
            if (ks > nchunks / 4) ks = nchunks / 4;             // at least 4 patches per block
            if (ctx->dbg_wgrad_ks > 0) ks = ctx->dbg_wgrad_ks;
            if (ks > nchunks) ks = nchunks;
            if (sub == 2) ks &= ~1ll;
            const bool sub2 = sub == 2 && ks >= 2;
            if (ks < 1) ks = 1;
            q.ksplit = (int)ks;
            q.atomic = ks > (sub2 ? 2 : 1) || acc_mode;
            const size_t g_elems = (size_t)Cout_real * Cin_real * d->KT * 9;
            if (q.atomic && !acc_mode && hipMemsetAsync(Gt, 0, g_elems * sizeof(float), st) != hipSuccess)
                DAT_FAIL(ctx, DAT_ERR_LAUNCH, "conv3d_wgrad: memset failed");
            q.zeros = (const char*)ctx->zeros;
            q.xcd = ctx->dbg_wgrad_xcd != 0;
            const bool ilv = ctx->dbg_wgrad_ilv != 0;
            const void* fn;
            if (sub2) {
                const size_t lds = 2 * W9D_STAGES * W9D_STAGE;
                const dim3 grid((unsigned)(tiles * ks / 2)), blk(2 * NT);
                fn = ilv ? (const void*)wgrad_dma9_kernel<2, 1> : (const void*)wgrad_dma9_kernel<2, 0>;
                if (dat_ensure_lds(ctx, fn, lds) != DAT_OK) return DAT_ERR_LAUNCH;
                if (ilv) hipLaunchKernelGGL((wgrad_dma9_kernel<2, 1>), grid, blk, lds, st, q);
                else hipLaunchKernelGGL((wgrad_dma9_kernel<2, 0>), grid, blk, lds, st, q);
            } else if (ctx->dbg_wgrad_dma) {   // operands by LDS-DMA into three stages (DAT_WGRAD_DMA, default 1)
                const size_t lds = W9D_STAGES * W9D_STAGE;
                const dim3 grid((unsigned)(tiles * ks)), blk(NT);
                fn = ilv ? (const void*)wgrad_dma9_kernel<1, 1> : (const void*)wgrad_dma9_kernel<1, 0>;
                if (dat_ensure_lds(ctx, fn, lds) != DAT_OK) return DAT_ERR_LAUNCH;
                if (ilv) hipLaunchKernelGGL((wgrad_dma9_kernel<1, 1>), grid, blk, lds, st, q);
                else hipLaunchKernelGGL((wgrad_dma9_kernel<1, 0>), grid, blk, lds, st, q);
            } else {
                if (dat_ensure_lds(ctx, (const void*)wgrad_direct9_kernel, 2 * W9_STAGE) != DAT_OK) return DAT_ERR_LAUNCH;
                hipLaunchKernelGGL(wgrad_direct9_kernel, dim3((unsigned)(tiles * ks)), dim3(NT), 2 * W9_STAGE, st, q);
            }
            if (!acc_mode)
                hipLaunchKernelGGL(wgrad_finish_kernel, dim3(grid_for((long long)g_elems, 256)), dim3(256), 0, st, (const float*)Gt, scale, dW,
                                   Cout_real, Cin_real, d->KT * 9);
            DAT_CHECK_LAUNCH(ctx, "conv3d_wgrad direct9");
            return DAT_OK;
        }
        if (pw_eligible(ctx, d)) {
            // pointwise layers (1 x 1 x 1 convs, FC; stride 1 | 2): eight-wave blocks, 64 K accumulators per tile (wgrad_pw_kernel)
            PwPlan pl;
            if (int rc = pw_plan(ctx, d, x, g, g_cstride, Cin_real, Cout_real, Gt, &pl)) return rc;
            const int ncu = dat_conv::ctx_num_cu(ctx);
            long long ks = pl.tiles >= ncu ? 1 : ncu / pl.tiles;        // one block per CU
            if (ks > pl.nchunks / 4) ks = pl.nchunks / 4;               // at least 4 chunks per block
            if (ctx->dbg_wgrad_ks > 0) ks = ctx->dbg_wgrad_ks;
            if (ks > pl.nchunks) ks = pl.nchunks;
            if (ks < 1) ks = 1;
            pl.it.ksplit = (int)ks;
            pl.it.atomic = ks > 1 || acc_mode;
            const size_t g_elems = (size_t)Cout_real * Cin_real;
            if (ks > 1 && !acc_mode && hipMemsetAsync(Gt, 0, g_elems * sizeof(float), st) != hipSuccess)
                DAT_FAIL(ctx, DAT_ERR_LAUNCH, "conv3d_wgrad: memset failed");
            if (int rc = pw_launch(ctx, st, &pl, 1)) return rc;
            if (!acc_mode)
                hipLaunchKernelGGL(wgrad_finish_kernel, dim3(grid_for((long long)g_elems, 256)), dim3(256), 0, st, (const float*)Gt, scale, dW,
                                   Cout_real, Cin_real, 1);
            DAT_CHECK_LAUNCH(ctx, "conv3d_wgrad pointwise");
            return DAT_OK;
        }
        WgradDirectParams wp;
        memset(&wp, 0, sizeof(wp));
        wp.g = (const char*)g; wp.x = (const char*)x; wp.G = Gt;
        wp.Cout = Cout_real; wp.Cin = Cin_real; wp.g_cs = g_cstride; wp.x_cs = d->Cin;
        wp.T = d->T; wp.H = d->H; wp.W = d->W; wp.Ho = Ho; wp.Wo = Wo; wp.stride = s;
        wp.KT = d->KT; wp.KH = d->KH; wp.KW = d->KW; wp.pt = d->pad_t; wp.ph = d->pad_h; wp.pw = d->pad_w;
        wp.ntaps = d->KT * d->KH * d->KW;
        wp.n_co_tiles = (Cout_real + 127) / 128; wp.n_ci_tiles = (Cin_real + 127) / 128;
        wp.how = (unsigned)(Ho * Wo);
        wp.wo_magic = Wo == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)Wo - 1) / (unsigned)Wo);
        wp.p_begin = 0; wp.p_end = (unsigned)npos_out;
        if (d->out_tn > 0 && clips == 1) {   // g is non-zero only in frames [out_t0, out_t0 + out_tn) of the clip
            DAT_ENFORCE(ctx, d->out_t0 >= 0 && d->out_t0 + d->out_tn <= d->T, "conv3d_wgrad: gradient frames [%d, %d) outside T %d",
                        d->out_t0, d->out_t0 + d->out_tn, d->T);
            wp.p_begin = (unsigned)d->out_t0 * wp.how; wp.p_end = (unsigned)(d->out_t0 + d->out_tn) * wp.how;
        }
        // (__umulhi(p, ceil(2^32 / n)) == p / n needs p * n < 2^32 or so: checked exactly for the ranges used)
        const long long tiles = (long long)wp.ntaps * wp.n_co_tiles * wp.n_ci_tiles;
        const long long nchunks = ((long long)(wp.p_end - wp.p_begin) + 63) / 64;
        long long ks = 1024 / tiles;                        // ~2 rounds of the 512 resident blocks; every split costs a partial tile
        if (ks > nchunks / 8) ks = nchunks / 8;             // at least 8 chunks per block
        if (ks < 1) ks = 1;
        wp.ksplit = (int)ks;
        wp.atomic = ks > 1 || acc_mode;
        const size_t g_elems = (size_t)Cout_real * Cin_real * wp.ntaps;
        if (ks > 1 && !acc_mode && hipMemsetAsync(Gt, 0, g_elems * sizeof(float), st) != hipSuccess)
            DAT_FAIL(ctx, DAT_ERR_LAUNCH, "conv3d_wgrad: memset failed");
        if (dat_ensure_lds(ctx, (const void*)wgrad_direct_kernel, 4 * WD_TILE) != DAT_OK) return DAT_ERR_LAUNCH;
        hipLaunchKernelGGL(wgrad_direct_kernel, dim3((unsigned)(tiles * ks)), dim3(NT), 4 * WD_TILE, st, wp);
        if (!acc_mode)
            hipLaunchKernelGGL(wgrad_finish_kernel, dim3(grid_for((long long)g_elems, 256)), dim3(256), 0, st, (const float*)Gt, scale, dW,
                               Cout_real, Cin_real, wp.ntaps);
        DAT_CHECK_LAUNCH(ctx, "conv3d_wgrad direct");
        return DAT_OK;
    }
    char* gT = (char*)workspace;
    char* xT = gT + (((size_t)Cout_real * Qa * es + 255) & ~(size_t)255);
    const size_t plane_bytes = (size_t)Cin_real * Qa * es;
    float* Gt = (float*)(xT + (((size_t)s * s * nc * plane_bytes + 255) & ~(size_t)255));

    PackParams pp;
    pp.clips = clips; pp.Tq = Tq; pp.Hq = Hq; pp.Wq = Wq; pp.Qtot = Qtot; pp.Qa = Qa;
    auto launch_pack = [&](const PackParams& q) {
        dim3 grid((unsigned)((Qa + 63) / 64), (unsigned)((q.C + 63) / 64));
        if (d->dtype == DAT_BF16) hipLaunchKernelGGL(cq_pack_kernel<DAT_BF16>, grid, dim3(256), 0, st, q);
        else hipLaunchKernelGGL(cq_pack_kernel<DAT_F32>, grid, dim3(256), 0, st, q);
    };
    // gT: the output-gradient grid, no shift, zero beyond (T, Ho, Wo)
    pp.src = (const char*)g; pp.dst = gT; pp.T = d->T; pp.Hs = Ho; pp.Ws = Wo; pp.cs = g_cstride; pp.C = Cout_real;
    pp.st = 1; pp.ot = 0; pp.oy = 0; pp.ox = 0; pp.shift = 0;
    launch_pack(pp);
    // xT planes (stride parity) x copies (element shift)
    pp.src = (const char*)x; pp.T = d->T; pp.Hs = d->H; pp.Ws = d->W; pp.cs = d->Cin; pp.C = Cin_real; pp.st = s;
    pp.ot = d->pad_t;
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px)
            for (int c = 0; c < nc; ++c) {
                pp.dst = xT + ((size_t)((py * s + px) * nc + c)) * plane_bytes;
                pp.oy = py - d->pad_h; pp.ox = px - d->pad_w; pp.shift = c;
                launch_pack(pp);
            }
    DAT_CHECK_LAUNCH(ctx, "conv3d_wgrad pack");

    WgradParams wp;
    memset(&wp, 0, sizeof(wp));
    wp.gT = gT; wp.xT = xT; wp.G = Gt; wp.Cout = Cout_real; wp.Cin = Cin_real;
    wp.ntaps = d->KT * d->KH * d->KW; wp.Qa = Qa;
    const int ck = d->dtype == DAT_BF16 ? 64 : 32;
    wp.K = (Qtot + ck - 1) / ck * ck;
    wp.k_begin = 0;
    // d->out_t0 / out_tn (optional): g is non-zero only in frames [out_t0, out_t0 + out_tn) of the clip (a key-frame gradient
    // embedded in its temporal window): reduce over those frames' positions only
    if (d->out_tn > 0 && clips == 1) {
        DAT_ENFORCE(ctx, d->out_t0 >= 0 && d->out_t0 + d->out_tn <= d->T, "conv3d_wgrad: gradient frames [%d, %d) outside T %d",
                    d->out_t0, d->out_t0 + d->out_tn, d->T);
        const long long fq = (long long)Hq * Wq;
        const long long b0 = (long long)d->out_t0 * fq / ck * ck;
        const long long e0 = ((long long)(d->out_t0 + d->out_tn) * fq + ck - 1) / ck * ck;
        wp.k_begin = b0;
        wp.K = (e0 < wp.K ? e0 : wp.K) - b0;
    }
    wp.n_co_tiles = (Cout_real + 127) / 128; wp.n_ci_tiles = (Cin_real + 127) / 128;
    for (int kt = 0; kt < d->KT; ++kt)
        for (int kh = 0; kh < d->KH; ++kh)
            for (int kw = 0; kw < d->KW; ++kw) {
                const int tap = (kt * d->KH + kh) * d->KW + kw;
                long long off = ((long long)kt * Hq + kh / s) * Wq + kw / s;
                int copy = 0;
                if (nc == 2 && (off & 1)) { copy = 1; off -= 1; }
                wp.tap_off[tap] = off;
                wp.tap_base[tap] = (long long)(((kh % s) * s + (kw % s)) * nc + copy) * (long long)plane_bytes;
            }
    const long long tiles = (long long)wp.ntaps * wp.n_co_tiles * wp.n_ci_tiles;
    const long long ksteps = wp.K / ck;
    long long ks = 1024 / tiles;                        // ~2 rounds of the 512 resident blocks; every split costs a partial tile
    if (ks > ksteps / 16) ks = ksteps / 16;             // at least 16 K-steps per block
    if (ks < 1) ks = 1;
    wp.ksplit = (int)ks;
    const size_t g_elems = (size_t)Cout_real * Cin_real * wp.ntaps;
    if (ks > 1 && hipMemsetAsync(Gt, 0, g_elems * sizeof(float), st) != hipSuccess)
        DAT_FAIL(ctx, DAT_ERR_LAUNCH, "conv3d_wgrad: memset failed");
    const unsigned nblocks = (unsigned)(tiles * ks);
    if (d->dtype == DAT_BF16) hipLaunchKernelGGL(wgrad_gemm_kernel<DAT_BF16>, dim3(nblocks), dim3(NT), 0, st, wp);
    else hipLaunchKernelGGL(wgrad_gemm_kernel<DAT_F32>, dim3(nblocks), dim3(NT), 0, st, wp);
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(grid_for((long long)g_elems, 256)), dim3(256), 0, st, (const float*)Gt, scale, dW,
                       Cout_real, Cin_real, wp.ntaps);
    DAT_CHECK_LAUNCH(ctx, "conv3d_wgrad gemm");
    return DAT_OK;
}

int dat_conv3d_wgrad(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x, const void* g, int g_cstride,
                     int Cin_real, int Cout_real, const float* scale, void* workspace, float* dW) {
    return wgrad_impl(ctx, s, d, x, g, g_cstride, Cin_real, Cout_real, scale, workspace, dW, 0);
}

int dat_conv3d_wgrad_acc_supported(dat_ctx* ctx, const dat_conv_desc* d, int g_cstride) {
    return ctx && d && wgrad_direct_eligible(ctx, d, g_cstride) ? 1 : 0;
}

int dat_conv3d_wgrad_acc(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x, const void* g, int g_cstride,
                         int Cin_real, int Cout_real, float* Gt) {
    return wgrad_impl(ctx, s, d, x, g, g_cstride, Cin_real, Cout_real, nullptr, Gt, nullptr, 1);
}

// n deferred-finish weight gradients in one call (same result as n dat_conv3d_wgrad_acc calls): the pointwise layers among them share
// grouped launches of wgrad_pw_kernel -- ~2 blocks per CU over ALL of them, every block about the same number of 32-position chunks --,
// the others run one by one.
int dat_conv3d_wgrad_acc_batch(dat_ctx* ctx, dat_stream s, const dat_wgrad_job* jobs, int n) {
    DAT_ENFORCE(ctx, jobs || n == 0, "conv3d_wgrad_acc_batch: null argument");
    std::vector<PwPlan> pls;
    for (int i = 0; i < n; ++i) {
        const dat_wgrad_job& j = jobs[i];
        DAT_ENFORCE(ctx, j.desc && j.x && j.g && j.Gt, "conv3d_wgrad_acc_batch: null member in job %d", i);
        if (wgrad_direct_eligible(ctx, j.desc, j.g_cstride) && pw_eligible(ctx, j.desc)) {
            PwPlan pl;
            if (int rc = pw_plan(ctx, j.desc, j.x, j.g, j.g_cstride, j.Cin_real, j.Cout_real, j.Gt, &pl)) return rc;
            pls.push_back(pl);
        } else if (int rc = wgrad_impl(ctx, s, j.desc, j.x, j.g, j.g_cstride, j.Cin_real, j.Cout_real, nullptr, j.Gt, nullptr, 1)) {
            return rc;
        }
    }
    if (pls.empty()) return DAT_OK;
    const int ncu = dat_conv::ctx_num_cu(ctx);
    long long total = 0;
    for (const PwPlan& pl : pls) total += pl.tiles * pl.nchunks;
    const long long target = (pls.size() == 1 ? 1 : 2) * (long long)ncu;           // blocks in the grid(s)
    long long cpb = (total + target - 1) / target;                                  // chunks per block
    if (cpb < 4) cpb = 4;
    for (PwPlan& pl : pls) {
        long long ks = (pl.nchunks + cpb - 1) / cpb;
        if (ctx->dbg_wgrad_ks > 0) ks = ctx->dbg_wgrad_ks;
        if (ks > pl.nchunks) ks = pl.nchunks;
        if (ks < 1) ks = 1;
        pl.it.ksplit = (int)ks;
        pl.it.atomic = 1;
    }
    return pw_launch(ctx, (hipStream_t)s, pls.data(), (int)pls.size());
}

// dW[co][ci][tap] (+)= scale[co] * Gt[tap][co][ci] for a table of layers in ONE launch (blocks of 256 threads x 8 elements; an item owns
// the blocks [block0, block0 + ceil(elems / 2048))): the same arithmetic per element as wgrad_finish_kernel
__global__ __launch_bounds__(256) void wgrad_finish_batch_kernel(const dat_wfinish_item* __restrict__ items, int n) {
    const long long b = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    const dat_wfinish_item it = items[lo];
    const long long total = (long long)it.Cout * it.Cin * it.ntaps;
    const long long i0 = (b - it.block0) * 2048 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long long i = i0 + u * 256;
        if (i >= total) break;
        const int tap = (int)(i % it.ntaps);
        const long long r = i / it.ntaps;
        const int ci = (int)(r % it.Cin), co = (int)(r / it.Cin);
        const float v = it.Gt[((size_t)tap * it.Cout + co) * it.Cin + ci];
        const float o = it.scale ? v * it.scale[co] : v;
        it.dW[i] = it.accumulate ? it.dW[i] + o : o;
    }
}

int dat_wgrad_finish_batch(dat_ctx* ctx, dat_stream s, const dat_wfinish_item* items_dev, int n, long long total_blocks) {
    DAT_ENFORCE(ctx, items_dev && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "wgrad_finish_batch: bad argument");
    hipLaunchKernelGGL(wgrad_finish_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)s, items_dev, n);
    DAT_CHECK_LAUNCH(ctx, "wgrad_finish_batch");
    return DAT_OK;
}

int dat_relu_bias_bwd(dat_ctx* ctx, dat_stream s, int dtype, const void* dy, const void* dy2, const void* y, void* g,
                      float* dbias, long long npos, int C, int cstride, int relu) {
    DAT_ENFORCE(ctx, dy && (y || !relu) && (g || (!relu && !dy2 && dbias && C == cstride)), "relu_bias_bwd: null argument");
    const int nq = cstride / 4;
    const int block = nq > 256 ? nq : 256;              // a thread keeps ONE channel quad: block % nq == 0
    DAT_ENFORCE(ctx, cstride % 4 == 0 && block <= 1024 && block % nq == 0,
                "relu_bias_bwd: channel stride %d must be 4 * (a divisor of 256, or 512 / 1024)", cstride);
    if (npos == 0) return DAT_OK;
    long long blocks = (npos * nq + block - 1) / block;
    const long long cap = dbias ? 2048 : 4096;          // (bias-reducing launches write one partial row per block)
    if (blocks > cap) blocks = cap;
    float* partial = nullptr;
    if (dbias) {
        const int rc = dat_ensure_ws(ctx, (size_t)blocks * cstride * sizeof(float));
        if (rc != DAT_OK) return rc;
        partial = (float*)ctx->ws;
    }
    if (dtype == DAT_BF16)
        hipLaunchKernelGGL(relu_bwd_kernel<DAT_BF16>, dim3((unsigned)blocks), dim3(block), 0, (hipStream_t)s, dy, dy2, y, g, dbias, partial, npos, C, cstride, relu);
    else
        hipLaunchKernelGGL(relu_bwd_kernel<DAT_F32>, dim3((unsigned)blocks), dim3(block), 0, (hipStream_t)s, dy, dy2, y, g, dbias, partial, npos, C, cstride, relu);
    if (dbias)
        hipLaunchKernelGGL(bias_partial_reduce_kernel, dim3((unsigned)((C + 31) / 32)), dim3(32 * BPR_LANES), 0, (hipStream_t)s, (const float*)partial, (int)blocks,
                           cstride, C, dbias);
    DAT_CHECK_LAUNCH(ctx, "relu_bias_bwd");
    return DAT_OK;
}

int dat_zero_insert2x(dat_ctx* ctx, dat_stream s, int dtype, const void* src, void* dst, int frames, int Hs, int Ws, int Hd,
                      int Wd, int cstride) {
    DAT_ENFORCE(ctx, src && dst && cstride % 4 == 0, "zero_insert2x: bad argument");
    DAT_ENFORCE(ctx, Hd >= 2 * Hs - 1 && Wd >= 2 * Ws - 1, "zero_insert2x: %dx%d does not hold 2x of %dx%d", Hd, Wd, Hs, Ws);
    const long long n = (long long)frames * Hd * Wd * (cstride / 4);
    if (dtype == DAT_BF16)
        hipLaunchKernelGGL(zero_insert2x_kernel<DAT_BF16>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)s, src, dst, frames, Hs, Ws, Hd, Wd, cstride);
    else
        hipLaunchKernelGGL(zero_insert2x_kernel<DAT_F32>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)s, src, dst, frames, Hs, Ws, Hd, Wd, cstride);
    DAT_CHECK_LAUNCH(ctx, "zero_insert2x");
    return DAT_OK;
}

int dat_upsample2x_bwd(dat_ctx* ctx, dat_stream s, int dtype, const void* g, void* dtop, int frames, int Ht, int Wt,
                       int cstride, int accumulate) {
    DAT_ENFORCE(ctx, g && dtop, "upsample2x_bwd: null argument");
    const long long n = (long long)frames * Ht * Wt * cstride;
    if (dtype == DAT_BF16)
        hipLaunchKernelGGL(upsample2x_bwd_kernel<DAT_BF16>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)s, g, dtop, frames, Ht, Wt, cstride, accumulate);
    else
        hipLaunchKernelGGL(upsample2x_bwd_kernel<DAT_F32>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)s, g, dtop, frames, Ht, Wt, cstride, accumulate);
    DAT_CHECK_LAUNCH(ctx, "upsample2x_bwd");
    return DAT_OK;
}

int dat_roi_align_bwd(dat_ctx* ctx, dat_stream s, int dtype, float* const* dfeat_levels, const int* Hs, const int* Ws,
                      const float* scales, int n_levels, int k_min, float canon_scale, int canon_level, int T, int C,
                      const float* rois, int R, int Tr, int t0, int pooled, int sampling_ratio, const void* dout) {
    DAT_ENFORCE(ctx, dfeat_levels && Hs && Ws && scales && rois && dout, "roi_align_bwd: null argument");
    DAT_ENFORCE(ctx, n_levels >= 1 && n_levels <= 4, "roi_align_bwd: n_levels %d must be 1..4", n_levels);
    if (R == 0) return DAT_OK;
    RoiBwdParams p;
    for (int i = 0; i < n_levels; ++i) { p.dfeat[i] = dfeat_levels[i]; p.H[i] = Hs[i]; p.W[i] = Ws[i]; p.scale[i] = scales[i]; }
    p.n_levels = n_levels; p.k_min = k_min; p.canon_level = canon_level; p.canon_scale = canon_scale;
    p.T = T; p.C = C; p.rois = rois; p.R = R; p.Tr = Tr; p.t0 = t0; p.pooled = pooled; p.sampling = sampling_ratio;
    p.dout = (const char*)dout;
    p.fold = ctx->dbg_roi_fold;
    const long long ncell = (long long)R * Tr * pooled * pooled;
    long long blocks = (ncell + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (dtype == DAT_BF16) hipLaunchKernelGGL(roi_align_bwd_kernel<DAT_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p);
    else hipLaunchKernelGGL(roi_align_bwd_kernel<DAT_F32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p);
    DAT_CHECK_LAUNCH(ctx, "roi_align_bwd");
    return DAT_OK;
}

int dat_kps_finalize_bwd(dat_ctx* ctx, dat_stream s, int dtype, const float* dout, int R, int Tr, int S, int cs, int K, int up,
                         void* dsub) {
    DAT_ENFORCE(ctx, dout && dsub && up >= 2 && up % 2 == 0 && 4 * K <= cs, "kps_finalize_bwd: bad argument");
    if (R == 0) return DAT_OK;
    const long long n = (long long)R * Tr * S * S * cs;
    if (dtype == DAT_BF16)
        hipLaunchKernelGGL(kps_finalize_bwd_kernel<DAT_BF16>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)s, dout, R, Tr, S, cs, K, up, dsub);
    else
        hipLaunchKernelGGL(kps_finalize_bwd_kernel<DAT_F32>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)s, dout, R, Tr, S, cs, K, up, dsub);
    DAT_CHECK_LAUNCH(ctx, "kps_finalize_bwd");
    return DAT_OK;
}

int dat_sgd_momentum(dat_ctx* ctx, dat_stream s, float* w, float* v, const float* grad, long long n, float lr, float momentum,
                     float weight_decay, int is_bias) {
    DAT_ENFORCE(ctx, w && v && grad, "sgd_momentum: null argument");
    if (n == 0) return DAT_OK;
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)s, w, v, grad, n, lr, momentum, weight_decay, is_bias);
    DAT_CHECK_LAUNCH(ctx, "sgd_momentum");
    return DAT_OK;
}

}  // extern "C"
