// The special-cased conv kernels of round 2 ("Round 2") and their launchers: weights-stationary persistent
// kernels for the 64-channel layers (res2's 3x3 convs, the FPN P2 lateral) and the big-tile 3x3 kernel.  The dispatcher that
// chooses between them and the generic kernel is dat_conv3d_fwd (conv3d_igemm.hip).
#include "conv_internal.h"

using namespace dat_conv;

namespace {

// ---- weights-stationary 3x3, 64 -> 64 channels (the res2 stage: ResNet3D.py:266-272 at 1/4 resolution) -------------------------
// The generic kernel streams a layer's weights to every block; with 64 output channels a block has only 2 row blocks of MFMA work
// per weight fragment, so the 72 KB of weights per 128 positions are the traffic that bounds it (each MFMA needs a fresh 1-KiB
// fragment through the CU's 64 B/clk L1: 0.12 ms per layer in the network, 310 TFLOP/s, against a 0.03 ms HBM floor).  Here the
// WHOLE weight tensor (9 taps x 64 x 64 bf16 = 72 fragments of 16 B per lane) lives in each wave's registers for the life of a
// PERSISTENT block (one block per CU, one wave per SIMD, 512-register budget), and the block walks over output tiles:
//   * a tile is 256 positions (8 x 32 or 16 x 16); its (TH+2) x (TW+2) x 128-B input patch is double-buffered in LDS by LDS-DMA
//     -- the next tile's patch is requested before the current tile's MFMAs, so HBM latency never shows;
//   * per tile a wave runs 9 taps x 4 k-slices x (2 B-fragment ds_read_b128 + 4 MFMAs): the only operand traffic is 72 LDS reads
//     for 144 MFMAs (every B fragment feeds both 32-channel row blocks), no weight traffic at all, ONE barrier per tile;
//   * the epilogue (affine, residual, ReLU, 16-byte channel-contiguous stores) goes through a per-wave LDS slice as in the
//     generic kernel.
struct Ws64Params {
    const char* x;
    const char* w;              // MFMA-fragment order (pack_weights*, frag = 1): [tap][32-row block][k-slice][lane][16 B]
    const float* scale;
    const float* bias;
    const char* res;
    char* y;
    const char* zeros;
    int frames, H, W, out_cs, relu, res_mode;
    int tiles_h, tiles_w, ntiles;
    int ablate;                 // DEBUG (DAT_CONV_ABLATE): 1 skip the patch loads after the first, 4 skip the stores (and residual loads)
};

template <int TWL>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_c64_ws_kernel(const Ws64Params p) {
    constexpr int TW = 1 << TWL, TH = 256 >> TWL, PW = TW + 2, PH = TH + 2;
    constexpr int NPIX = PH * PW, NPIECE = (NPIX * 8 + 63) / 64, PBYTES = NPIECE * 1024, UMAX = (NPIECE + 3) / 4;
    constexpr int EPITCH = 64 * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, n = lane & 31;
    char* const est = smem + 2 * PBYTES + wave * (32 * EPITCH);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    // ---- the layer's weights: 72 fragments, resident: taps 0-2 in VGPRs, taps 3-8 in AGPRs (with the 64 accumulators: all 256) ----
    // The MFMAs below are inline assembly for exactly this reason: gfx950 MFMAs read their A operand from either register file, but
    // the compiler only ever used the AGPR half as spill space (4 v_accvgpr_read per fragment and tile) and, out of VGPRs,
    // serialised every ds_read behind an lgkmcnt(0).  The "v" / "a" constraints pin each fragment to its file.
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    constexpr int VT = 3;                                   // taps held in VGPRs
    u32x4_t wv[VT][2][4], wg[9 - VT][2][4];
    {
        const char* wl = p.w + lane * 16;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4_t v = *(const u32x4_t*)(wl + ((tp * 2 + mb) * 4 + ks) * 1024);
                    if (tp < VT) wv[tp][mb][ks] = v; else wg[tp - VT][mb][ks] = v;
                }
    }
#define WS_MFMA0_V(ACC_, A_, B_) asm volatile(DAT_MFMA16_OP " %0, %1, %2, 0" : "=&a"(ACC_) : "v"(A_), "v"(B_) : "memory")
#define WS_MFMA_V(ACC_, A_, B_) asm volatile(DAT_MFMA16_OP " %0, %1, %2, %0" : "+a"(ACC_) : "v"(A_), "v"(B_) : "memory")
#define WS_MFMA_A(ACC_, A_, B_) asm volatile(DAT_MFMA16_OP " %0, %1, %2, %0" : "+a"(ACC_) : "a"(A_), "v"(B_) : "memory")
    // ---- swizzled LDS address (k-slice 0) of the B fragment of every (tap, position sub-tile), two per register ----
    // sub-tile j of this wave: 8 x 32 tiles: output row 2*wave + j, column n; 16 x 16 tiles: rows 4*wave + 2*j + (n >> 4), column n & 15
    unsigned qp[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        unsigned q = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = TWL == 5 ? 2 * wave + j : 4 * wave + 2 * j + (n >> 4), c = TWL == 5 ? n : (n & 15);
            const int row = (r + tp / 3) * PW + c + tp % 3;
            const int g = (row >> 1) & 7;
            const unsigned a16 = (unsigned)(row * PPITCH) + (unsigned)(((khalf ^ (g & 1)) << 4) | ((g >> 1) << 5));
            q |= a16 << (16 * j);
        }
        qp[tp] = q;
    }
    static_assert(PBYTES < 65536, "two 16-bit patch addresses per register");
    // ---- epilogue constants: this lane's 8 channels in the store phase ----
    const int sl_c = (lane & 7) * 8, sl_p = lane >> 3;
    float sc[8], bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = p.scale ? p.scale[sl_c + e] : 1.f;
        bi[e] = p.bias ? p.bias[sl_c + e] : 0.f;
    }
    const int tiles_per_frame = p.tiles_h * p.tiles_w;

    // patch of tile `tl` -> LDS buffer `b` (this wave's 1-KiB pieces wave, wave + 4, ...): lane-linear LDS-DMA image, the XOR swizzle
    // applied on the source side; halo pixels outside the frame (and the tail lanes of the last piece) fetch zeros
#define WS_DMA(TL_, B_)                                                                                                   \
    {                                                                                                                     \
        const int f_ = (TL_) / tiles_per_frame, r_ = (TL_) - f_ * tiles_per_frame;                                        \
        const int th_ = r_ / p.tiles_w, tw_ = r_ - th_ * p.tiles_w;                                                       \
        const int ih0_ = th_ * TH - 1, iw0_ = tw_ * TW - 1;                                                               \
        const char* xf_ = p.x + (size_t)f_ * p.H * p.W * 128;                                                             \
        _Pragma("unroll") for (int u_ = 0; u_ < UMAX; ++u_) {                                                             \
            const int piece_ = wave + 4 * u_;                                                                             \
            if (piece_ < NPIECE) {                                                                                        \
                const int it_ = piece_ * 64 + lane, row_ = it_ >> 3;                                                      \
                const int slot_ = (it_ ^ (row_ >> 1)) & 7;                                                                \
                const int prow_ = row_ / PW, pcol_ = row_ - prow_ * PW;                                                   \
                const int ih_ = ih0_ + prow_, iw_ = iw0_ + pcol_;                                                         \
                const bool ok_ = row_ < NPIX && (unsigned)ih_ < (unsigned)p.H && (unsigned)iw_ < (unsigned)p.W;           \
                const char* src_ = ok_ ? xf_ + ((unsigned)(ih_ * p.W + iw_) * 128u + (unsigned)(slot_ * 16)) : p.zeros;   \
                __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(smem + (B_) * PBYTES + piece_ * 1024), 16, 0, 0); \
            }                                                                                                             \
        }                                                                                                                 \
    }

    int tile = blockIdx.x;
    if (tile < p.ntiles) WS_DMA(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int buf = 0;
    for (; tile < p.ntiles; tile += gridDim.x, buf ^= 1) {
        // every wave has retired its DMA pieces of this tile (the vmcnt(0) before its last epilogue / above) and has left the other
        // buffer (its reads ended before that epilogue): one barrier publishes the patch and frees the other buffer
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < p.ntiles && !(p.ablate & 1)) WS_DMA(next, buf ^ 1);
        f32x16_t acc[2][2];
        const unsigned boff = (unsigned)buf * PBYTES;
        __builtin_amdgcn_s_setprio(1);
        // 36 steps (tap, k-slice) of 2 B-fragment reads + 4 MFMAs; the reads run two steps ahead through a 3-slot register ring (one
        // wave per SIMD: nothing else hides the LDS latency).  The statements are volatile asm with memory clobbers: program order.
        u32x4_t bq[3][2];
        unsigned a0 = 0, a1 = 0;
#define WS_READ(S_)                                                                                  \
        {                                                                                            \
            if ((S_) % 4 == 0) {                                                                     \
                unsigned q_ = qp[(S_) / 4];                                                          \
                asm volatile("" : "+v"(q_));   /* opaque per tile: keeps the 72 unpacked / xor-ed addresses out of registers */ \
                a0 = (q_ & 0xffffu) + boff; a1 = (q_ >> 16) + boff;                                  \
            }                                                                                        \
            asm volatile("ds_read_b128 %0, %1" : "=v"(bq[(S_) % 3][0]) : "v"(a0 ^ (unsigned)(((S_) % 4) << 5)) : "memory"); \
            asm volatile("ds_read_b128 %0, %1" : "=v"(bq[(S_) % 3][1]) : "v"(a1 ^ (unsigned)(((S_) % 4) << 5)) : "memory"); \
        }
        WS_READ(0);
        WS_READ(1);
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            if (st + 2 < 36) WS_READ(st + 2);
            // LDS returns in order: all but the reads of the (up to) two later steps have landed.  (The reads are asm too, so the
            // compiler's own waitcnt pass -- which waited for lgkmcnt(0) before every asm use of a pending ds_read -- stays out.)
            if (st + 2 < 36) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else if (st + 1 < 36) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int tp = st / 4, ks = st % 4;
            const u32x4_t b0 = bq[st % 3][0], b1 = bq[st % 3][1];
            if (st == 0) {
                WS_MFMA0_V(acc[0][0], wv[0][0][0], b0); WS_MFMA0_V(acc[1][0], wv[0][1][0], b0);
                WS_MFMA0_V(acc[0][1], wv[0][0][0], b1); WS_MFMA0_V(acc[1][1], wv[0][1][0], b1);
            } else if (tp < VT) {
                WS_MFMA_V(acc[0][0], wv[tp < VT ? tp : 0][0][ks], b0); WS_MFMA_V(acc[1][0], wv[tp < VT ? tp : 0][1][ks], b0);
                WS_MFMA_V(acc[0][1], wv[tp < VT ? tp : 0][0][ks], b1); WS_MFMA_V(acc[1][1], wv[tp < VT ? tp : 0][1][ks], b1);
            } else {
                WS_MFMA_A(acc[0][0], wg[tp >= VT ? tp - VT : 0][0][ks], b0); WS_MFMA_A(acc[1][0], wg[tp >= VT ? tp - VT : 0][1][ks], b0);
                WS_MFMA_A(acc[0][1], wg[tp >= VT ? tp - VT : 0][0][ks], b1); WS_MFMA_A(acc[1][1], wg[tp >= VT ? tp - VT : 0][1][ks], b1);
            }
        }
#undef WS_READ
        // (the compiler does not see MFMAs in the asm above: cover the XDL-write -> VALU-read wait states of the accumulators by hand)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's patch pieces (requested a whole tile of MFMAs ago)

        // ---- epilogue: per-wave LDS transpose, affine + residual + ReLU, 16-byte channel-contiguous stores ----
        // (interleaving it with the next tile's MFMAs was tried and dropped: vmcnt is one in-order counter, so the first residual
        //  load consumed -- and, through the compiler's LDS-DMA alias rule, the first LDS access of the epilogue -- would wait for the
        //  patch DMA issued just before it)
        const int f = tile / tiles_per_frame, rr = tile - f * tiles_per_frame;
        const int th_i = rr / p.tiles_w, tw_i = rr - th_i * p.tiles_w;
        const int oh0 = th_i * TH, ow0 = tw_i * TW;
        const size_t tile_pos = ((size_t)f * p.H + oh0) * p.W + ow0;
        char* const ybase = p.y + tile_pos * p.out_cs * 2;
        const char* const rbase = p.res + tile_pos * p.out_cs * 2;
        // residual rows one sub-tile ahead of their use (a load issued where it is consumed costs a memory round trip per store group)
        uint4 rq[2][4];
        auto res_fetch = [&](int j, uint4* dst) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pl = q * 8 + sl_p;
                const int ohl = TWL == 5 ? 2 * wave + j : 4 * wave + 2 * j + (pl >> 4), owl = TWL == 5 ? pl : (pl & 15);
                const bool live = oh0 + ohl < p.H && ow0 + owl < p.W && !(p.ablate & 4);
                dst[q] = *(const uint4*)(rbase + (live ? ((unsigned)(ohl * p.W + owl) * (unsigned)p.out_cs + (unsigned)sl_c) * 2u : 0u));
            }
        };
        if (p.res_mode) res_fetch(0, rq[0]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (p.res_mode && j == 0) res_fetch(1, rq[1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4*)(est + n * EPITCH + (i * 32 + g * 8 + khalf * 4) * 4) =
                        make_float4(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pl = q * 8 + sl_p;
                const float4 t0 = *(const float4*)(est + pl * EPITCH + sl_c * 4);
                const float4 t1 = *(const float4*)(est + pl * EPITCH + sl_c * 4 + 16);
                float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                const int ohl = TWL == 5 ? 2 * wave + j : 4 * wave + 2 * j + (pl >> 4), owl = TWL == 5 ? pl : (pl & 15);
                if (oh0 + ohl >= p.H || ow0 + owl >= p.W || (p.ablate & 4)) continue;
                const unsigned lpos = (unsigned)(ohl * p.W + owl);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bi[e];
                if (p.res_mode) {
                    const uint4 r = rq[j][q];
                    const uint32_t ru[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        v[2 * e2] += bf2f((uint16_t)(ru[e2] & 0xffff));
                        v[2 * e2 + 1] += bf2f((uint16_t)(ru[e2] >> 16));
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *(uint4*)(ybase + (lpos * (unsigned)p.out_cs + (unsigned)sl_c) * 2u) =
                    make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#undef WS_DMA
#undef WS_MFMA0_V
#undef WS_MFMA_V
#undef WS_MFMA_A
}

// ---- weights-stationary 1x1, 64 -> 256 channels (the FPN P2 lateral: FPN3D.py:111-134 on res2, 1/4 resolution) ------------------
// An HBM-bound layer: 66 MB in, 264 MB out (+ 66 MB of top-down map).  The generic kernel's 128-channel x 128-position blocks write
// every output position as four 128-byte pieces from different waves and blocks, at different times; measured with cold caches that
// costs 0.117 of the layer's 0.188 ms -- 2.3 TB/s of stores on a part that fills memory at 5.8 TB/s (tools/hbm_probe.py).  Here a
// wave owns ALL 256 channels of its 32 positions: the whole weight matrix (32 fragments, 128 VGPRs) stays in its registers, the
// input fragments come straight from global memory (32 contiguous bytes per lane pair; prefetched one tile ahead), and after a
// per-wave LDS transpose every store instruction writes two complete 512-byte output rows.  Waves never synchronise; the grid is
// persistent (one block per CU: the fp32 transpose slice is 33 KB per wave).
struct Pw256Params {
    const char* x;
    const char* w;              // MFMA-fragment order: [32-row block][k-slice][lane][16 B]
    const float* scale;
    const float* bias;
    const char* res;
    char* y;
    long long npos;             // frames * H * W
    int H, W, out_cs, relu, res_mode;
    int ntiles;                 // wave tiles of 32 positions
    unsigned hw, w_magic;       // H * W; ceil(2^32 / W): row = umulhi(position in frame, w_magic)
};

__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv1x1_k64_c256_ws_kernel(const Pw256Params p) {
    constexpr int EPITCH = 256 * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, n = lane & 31;
    char* const est = smem + wave * (32 * EPITCH);
    uint4 wa[8][4];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wa[mb][ks] = *(const uint4*)(p.w + (mb * 4 + ks) * 1024 + lane * 16);
    // store phase: lane = 8 channels of one of two positions
    const int sl_c = (lane & 31) * 8, sl_p = lane >> 5;
    float sc[8], bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = p.scale ? p.scale[sl_c + e] : 1.f;
        bi[e] = p.bias ? p.bias[sl_c + e] : 0.f;
    }
    const int nwaves = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    // B fragment (k-slice ks) of position pos: bytes [ks * 32 + khalf * 16, +16) of its 128-byte channel row
    uint4 bcur[4], bnext[4];
#define PW_LOAD(DST_, TILE_)                                                                                   \
    {                                                                                                          \
        const unsigned pos_ = min((unsigned)(TILE_) * 32u + (unsigned)n, (unsigned)p.npos - 1u);               \
        const char* src_ = p.x + (size_t)(pos_ * 128u + (unsigned)khalf * 16u);                                                   \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) DST_[ks_] = *(const uint4*)(src_ + ks_ * 32);      \
    }
    if (tile < p.ntiles) PW_LOAD(bcur, tile);
    for (; tile < p.ntiles; tile += nwaves) {
        const int next = tile + nwaves;
        if (next < p.ntiles) PW_LOAD(bnext, next);
        // 32-bit position arithmetic (the launcher checks the ranges); the frame / row split of the tile's first position is
        // wave-uniform, a lane only adds its offset (one conditional frame wrap: a frame has >= 32 positions)
        const unsigned pos0 = (unsigned)tile * 32u;
        const unsigned f0 = pos0 / p.hw;
        // the residual rows of this tile (16 store groups of 2 positions) are requested NOW: one wave per SIMD has nothing else to
        // hide their latency behind than its own MFMA and transpose phases
        uint4 rr[16];
        if (p.res_mode) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const unsigned pl = (unsigned)(q * 2 + sl_p);
                const unsigned pos = min(pos0 + pl, (unsigned)p.npos - 1u);
                unsigned rpos = pos;
                if (p.res_mode == 2) {       // nearest-2x up-sampled coarser map
                    unsigned fr = f0, rem = pos - f0 * p.hw;
                    if (rem >= p.hw) { rem -= p.hw; ++fr; }
                    const unsigned oh = __umulhi(rem, p.w_magic), ow = rem - oh * (unsigned)p.W;
                    rpos = (fr * (unsigned)(p.H >> 1) + (oh >> 1)) * (unsigned)(p.W >> 1) + (ow >> 1);
                }
                rr[q] = *(const uint4*)(p.res + (size_t)((rpos * (unsigned)p.out_cs + (unsigned)sl_c) * 2u));
            }
        }
        f32x16_t acc[8];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) Mma<DAT_BF16>::step(wa[mb][ks], bcur[ks], acc[mb]);
        // ---- epilogue: transpose through this wave's LDS slice, then two complete output rows per store instruction ----
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(est + n * EPITCH + (mb * 32 + g * 8 + khalf * 4) * 4) =
                    make_float4(acc[mb][g * 4 + 0], acc[mb][g * 4 + 1], acc[mb][g * 4 + 2], acc[mb][g * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int pl = q * 2 + sl_p;
            const float4 t0 = *(const float4*)(est + pl * EPITCH + sl_c * 4);
            const float4 t1 = *(const float4*)(est + pl * EPITCH + sl_c * 4 + 16);
            float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            const unsigned pos = pos0 + (unsigned)pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bi[e];
            if (p.res_mode) {
                const uint32_t ru[4] = {rr[q].x, rr[q].y, rr[q].z, rr[q].w};
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    v[2 * e2] += bf2f((uint16_t)(ru[e2] & 0xffff));
                    v[2 * e2 + 1] += bf2f((uint16_t)(ru[e2] >> 16));
                }
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (pos < (unsigned)p.npos)
                *(uint4*)(p.y + (size_t)((pos * (unsigned)p.out_cs + (unsigned)sl_c) * 2u)) =
                    make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bcur[ks] = bnext[ks];
    }
#undef PW_LOAD
}

// ---- weights-in-LDS 1x1 kernel: K <= 512 -> C <= 512 (x cout parts), stride 1 | 2 (round 3) --------------------------------------
// The bottleneck 1x1x1 convs of R-50's res2 / res3 (ResNet3D.py:21-55: 256 -> 64, 512 -> 128, 128 -> 512 + Sum, the stride-2 first
// convs and shortcuts, :89-101) and the P2 / P3 laterals (FPN3D.py:111-134) are HBM-bound layers over 10^5..10^6 positions that the
// generic kernel runs at ~2.4 TB/s of algorithmic traffic (profiles/r03/before_conv_layers_3d_r50_fpn3d.txt): short-lived blocks,
// one 16-KB patch in flight per block, dependent residual loads.  This is conv1x1_k64_c256_ws_kernel's scheme for any K / C whose
// weight matrix (of one cout part) fits 128 KB of LDS: a persistent block (one per CU) copies the fragment-order weights into LDS
// once; then every wave streams its own 32-position tiles without ever synchronising -- B fragments straight from global memory
// into registers (all K channels of a position: KC x 4 16-byte pieces per lane), the NEXT tile's fragments requested before this
// tile's MFMAs, A fragments by ds_read_b128 (one per MFMA: at the <= 25 % matrix utilisation of these layers that is <= 25 % of the
// LDS rate), and a per-wave LDS transpose of 32 channels x 32 positions at a time so that every store instruction writes 16
// complete 64-byte channel runs; the residual rows of transpose group g + 1 are requested before group g is processed.
// Layers whose weights exceed 128 KB are split into `nsplit` cout parts: block b serves part b % nsplit (the parts read the same
// input tiles at about the same time: second reads hit L2 / MALL).
struct PwLwParams {
    const char* x;
    const char* w;              // MFMA-fragment order: [64-channel chunk][32-row block][k-slice][lane][16 B]
    const float* scale;
    const float* bias;
    const char* res;
    const char* res2;           // res_mode 4: the addend
    char* y;
    unsigned npos;              // output positions: frames * Ho * Wo
    int Ho, Wo, H, W, stride;
    int in_cs, out_cs, cout;    // channel strides (elements), real output channels
    int relu, res_mode;
    int ntiles;                 // wave tiles of 32 positions
    unsigned how, wo_magic;     // Ho * Wo; ceil(2^32 / Wo)
    int nsplit, mb_total;       // cout parts; Cout_pad / 32
    int xcd;                    // XCD-aware block -> (part, stream) map (the launcher sets it when the grid is a multiple of 8 * nsplit)
};

// M4 = 1: the instantiation for res_mode 4 only (in-place sum + mask: two operand rows per store group) -- separate, so that the other modes
// keep their register allocation (the two-pass variants sit at 256 + 196 registers already)
template <int KC, int MBW, int NPASS, int M4 = 0>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv1x1_lw_kernel(const PwLwParams p) {
    constexpr int EPITCH = 32 * 4 + 16;                 // fp32 transpose row of one position: 32 channels + pad
    constexpr int MBP = MBW * NPASS;                    // 32-row blocks of a block's cout part
    constexpr int WBYTES = KC * MBP * 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, n = lane & 31;
    // block -> (cout part, tile stream).  The parts of a layer read the SAME input tiles; blocks go to the XCDs round-robin (block b -> XCD
    // b % 8), so with part = b % nsplit the readers of a tile sat on different XCDs and every part fetched the tile through its own L2 (res4
    // `branch2c`, four parts: 264 MB of input reads for 66 MB of input).  XCD-aware (p.xcd, grid a multiple of 8 x nsplit): the k-th block of an
    // XCD serves part k % nsplit of stream (k / nsplit) * 8 + xcd -- all parts of a stream on one L2, launched back to back.
    int part = blockIdx.x % p.nsplit, stream = blockIdx.x / p.nsplit;
    if (p.xcd) {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        part = k % p.nsplit;
        stream = (k / p.nsplit) * 8 + xcd;
    }
    char* const wl = smem;                              // [kc][MBP][ks][lane][16 B]
    float* const sbl = (float*)(smem + WBYTES);         // scale[MBP * 32], bias[MBP * 32] of the part's channels
    char* const est = smem + WBYTES + MBP * 32 * 8 + wave * (32 * EPITCH);
    const int c_part0 = part * MBP * 32;                // first output channel of the part
    // ---- the part's weights, scale and bias -> LDS (once per block).  No global load may sit inside the store phase of a tile:
    // vmcnt is one in-order counter for loads AND stores, so waiting for such a load waits for every store issued before it ----
    for (int i = tid; i < WBYTES / 16; i += NTHREADS) {
        const int frag = i >> 6, l16 = i & 63;          // fragment (kc, mb, ks) of the part, 16-byte piece
        const int ks = frag & 3, mbl = (frag >> 2) % MBP, kc = (frag >> 2) / MBP;
        const size_t src = ((((size_t)kc * p.mb_total + (size_t)part * MBP + mbl) * 4 + ks) * 64 + l16) * 16;
        *(uint4*)(wl + (size_t)i * 16) = *(const uint4*)(p.w + src);
    }
    for (int i = tid; i < MBP * 32; i += NTHREADS) {
        const int c = c_part0 + i;
        const bool ok = c < p.cout;                     // channels past the stored Cout stay the zeros their zero weight rows produce
        sbl[i] = ok ? (p.scale ? p.scale[c] : 1.f) : 0.f;
        sbl[MBP * 32 + i] = (ok && p.bias) ? p.bias[c] : 0.f;
    }
    __syncthreads();
    // store phase: lane = 8 channels (sq) of one of 16 positions per round
    const int sq = lane & 3, spl = lane >> 2;
    const int nwaves = (gridDim.x / p.nsplit) * 4;
    int tile = stream * 4 + wave;
    uint4 bcur[KC * 4], bnext[KC * 4];
#define LW_LOAD(DST_, TILE_)                                                                                          \
    {                                                                                                                 \
        const unsigned pos_ = min((unsigned)(TILE_) * 32u + (unsigned)n, p.npos - 1u);                                \
        unsigned ip_ = pos_;                                                                                          \
        if (p.stride == 2) {                                                                                          \
            const unsigned f_ = pos_ / p.how, rem_ = pos_ - f_ * p.how;                                               \
            const unsigned oh_ = __umulhi(rem_, p.wo_magic), ow_ = rem_ - oh_ * (unsigned)p.Wo;                       \
            ip_ = (f_ * (unsigned)p.H + 2u * oh_) * (unsigned)p.W + 2u * ow_;                                         \
        }                                                                                                             \
        const char* src_ = p.x + ((size_t)ip_ * (unsigned)p.in_cs + (unsigned)khalf * 8u) * 2u;                       \
        _Pragma("unroll") for (int i_ = 0; i_ < KC * 4; ++i_) DST_[i_] = *(const uint4*)(src_ + (i_ >> 2) * 128 + (i_ & 3) * 32); \
    }
    if (tile < p.ntiles) LW_LOAD(bcur, tile);
    for (; tile < p.ntiles; tile += nwaves) {
        const int next = tile + nwaves;
        if (next < p.ntiles) LW_LOAD(bnext, next);
        const unsigned pos0 = (unsigned)tile * 32u;
        bool live[2];
        // ALL residual rows of the tile are requested now, before its MFMAs: one wave per SIMD has nothing else to hide their latency
        // behind, and a load issued between the stores would make its wait a wait for those stores
        // mode 4 (SUM with the addend rows + MASK by the residual operand: two rows per store group) is fetched PER
        // PASS, at its start: the addend into rr[0], the mask into rr[1] (two-pass variants: no register left for four prefetched sets) or rm
        uint4 rr[NPASS][MBW][2];
        uint4 rm[NPASS == 1 ? MBW : 1][2];
        constexpr bool m4 = M4 != 0;                // (the launcher picks the instantiation by p.res_mode)
        const char* rb[2];
        const char* ab[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned pos = pos0 + (unsigned)(r * 16 + spl);
            live[r] = pos < p.npos;
            if (m4) {
                const unsigned pc = min(pos, p.npos - 1u);
                rb[r] = p.res + ((size_t)pc * (unsigned)p.out_cs) * 2u;
                ab[r] = p.res2 + ((size_t)pc * (unsigned)p.out_cs) * 2u;
            } else if (p.res_mode) {
                const unsigned pc = min(pos, p.npos - 1u);
                unsigned rpos = pc;
                if (p.res_mode == 2) {       // nearest-2x up-sampled coarser map (FPN top-down, FPN3D.py:186-222)
                    const unsigned fr = pc / p.how, rem = pc - fr * p.how;
                    const unsigned oh = __umulhi(rem, p.wo_magic), ow = rem - oh * (unsigned)p.Wo;
                    rpos = (fr * (unsigned)(p.Ho >> 1) + (oh >> 1)) * (unsigned)(p.Wo >> 1) + (ow >> 1);
                }
                const char* rbl = p.res + ((size_t)rpos * (unsigned)p.out_cs) * 2u;
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) {
                        const unsigned c = (unsigned)min(c_part0 + (pass * MBW + mb) * 32 + sq * 8, p.out_cs - 8);
                        rr[pass][mb][r] = *(const uint4*)(rbl + c * 2u);
                    }
            }
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (m4) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) {
                        const unsigned c = (unsigned)min(c_part0 + (pass * MBW + mb) * 32 + sq * 8, p.out_cs - 8);
                        rr[0][mb][r] = *(const uint4*)(ab[r] + c * 2u);
                        if (NPASS == 1) rm[NPASS == 1 ? mb : 0][r] = *(const uint4*)(rb[r] + c * 2u);
                        else rr[NPASS - 1][mb][r] = *(const uint4*)(rb[r] + c * 2u);
                    }
            }
            f32x16_t acc[MBW];
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
            const char* wp = wl + (size_t)(pass * MBW) * 4096 + lane * 16;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) {
                        const uint4 a = *(const uint4*)(wp + ((size_t)(kc * MBP + mb) * 4 + ks) * 1024);
                        Mma<DAT_BF16>::step(a, bcur[kc * 4 + ks], acc[mb]);
                    }
            // ---- epilogue: 32 channels x 32 positions at a time through this wave's LDS slice; LDS and stores only ----
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) {
                const int cl = (pass * MBW + mb) * 32 + sq * 8;    // this lane's 8 channels inside the part
                const int c0 = c_part0 + cl;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4*)(est + n * EPITCH + (g * 8 + khalf * 4) * 4) =
                        make_float4(acc[mb][g * 4 + 0], acc[mb][g * 4 + 1], acc[mb][g * 4 + 2], acc[mb][g * 4 + 3]);
                __builtin_amdgcn_wave_barrier();
                const float4 s0 = *(const float4*)(sbl + cl), s1 = *(const float4*)(sbl + cl + 4);
                const float4 b0 = *(const float4*)(sbl + MBP * 32 + cl), b1 = *(const float4*)(sbl + MBP * 32 + cl + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const float bi[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int pl = r * 16 + spl;
                    const float4 t0 = *(const float4*)(est + pl * EPITCH + sq * 32);
                    const float4 t1 = *(const float4*)(est + pl * EPITCH + sq * 32 + 16);
                    float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bi[e];
                    if (m4) {
                        const uint4 qa = rr[0][mb][r], qm = NPASS == 1 ? rm[NPASS == 1 ? mb : 0][r] : rr[NPASS - 1][mb][r];
                        const uint32_t au[4] = {qa.x, qa.y, qa.z, qa.w}, mu[4] = {qm.x, qm.y, qm.z, qm.w};
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            v[2 * e2] = res_combine4(v[2 * e2], bf2f((uint16_t)(au[e2] & 0xffff)), bf2f((uint16_t)(mu[e2] & 0xffff)));
                            v[2 * e2 + 1] = res_combine4(v[2 * e2 + 1], bf2f((uint16_t)(au[e2] >> 16)), bf2f((uint16_t)(mu[e2] >> 16)));
                        }
                    } else if (p.res_mode) {            // modes 1 / 2: Sum (mode 3 layers stay on the generic kernel: dat_conv3d_fwd's dispatcher)
                        const uint4 q = rr[pass][mb][r];
                        const uint32_t ru[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            v[2 * e2] += bf2f((uint16_t)(ru[e2] & 0xffff));
                            v[2 * e2 + 1] += bf2f((uint16_t)(ru[e2] >> 16));
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (live[r] && c0 < p.cout) {
                        const unsigned pos = pos0 + (unsigned)pl;
                        *(uint4*)(p.y + ((size_t)pos * (unsigned)p.out_cs + (unsigned)c0) * 2u) =
                            make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#pragma unroll
        for (int i = 0; i < KC * 4; ++i) bcur[i] = bnext[i];
    }
#undef LW_LOAD
}

// ---- K-streaming 1x1 kernel (round 6): 1x1x1 convs whose weights do NOT fit LDS (K >= 512 with 256-channel cout blocks: R-50 / R-101 res4 /
// res5 `branch2a`, the res4 / res5 shortcuts, res5 `branch2c`, the P3 - P5 laterals, and the data gradients of the `branch2c` layers) --------
// These ran on the generic implicit-GEMM kernel, whose 128 x 128 tile re-fetches a 16-KB input patch AND 16 KB of weights per 2 MFLOP: with
// one tap there is nothing to reuse a patch for, and the layers sat at 1.5-2.4 TB/s of their compulsory traffic.  Here a block of EIGHT waves
// owns 256 positions x 256 output channels (64 K fp32 accumulators: a wave = 128 positions x 64 channels) and streams K in 64-channel
// chunks: the chunk of the input rows (32 KB, each row read from HBM exactly once per cout block) and the chunk of the packed weights (32 KB,
// L2-resident, already in MFMA-fragment order) go global -> LDS by LDS-DMA -- rows two chunks ahead into three buffers, weights one chunk
// ahead into two --, one barrier per chunk; a chunk is 32 MFMAs per wave for 24 fragment reads.  Blocks that share an input tile (the cout
// blocks of a layer) are neighbours in the grid.  The epilogue is the 1x1 kernels' per-wave LDS transpose: affine, residual (same-shape or
// nearest-2x top-down), ReLU, 16-byte channel-contiguous stores of complete 128-byte runs.
struct PwKsParams {
    const char* x;
    const char* w;              // MFMA-fragment order: [64-channel chunk][32-row block][k-slice][lane][16 B]
    const float* scale;
    const float* bias;
    const char* res;
    const char* res2;           // res_mode 4: the addend
    char* y;
    unsigned npos;              // output positions: frames * Ho * Wo
    int Ho, Wo, H, W, stride;
    int in_cs, out_cs, cout;
    int relu, res_mode;
    int ncb;                    // cout blocks of 256
    int kchunks, mb_total;      // Cin / 64; Cout_pad / 32
    unsigned how, wo_magic;     // Ho * Wo; ceil(2^32 / Wo)
    int ablate;                 // DEBUG (DAT_CONV_ABLATE): 1 no input-row copies after the first chunks, 2 no weight copies, 4 no MFMAs
    int xcd;                    // XCD-aware block order (more than one cout block)
};

constexpr int KS_THREADS = 512;
constexpr int KS_XBYTES = 256 * 128, KS_WBYTES = 8 * 4 * 1024;      // one K chunk of the input tile / of the weights of a cout block

__global__ __launch_bounds__(KS_THREADS) void conv1x1_ks_kernel(const PwKsParams p) {
    constexpr int EPITCH = 64 * 4 + 16;                 // fp32 transpose row of one position: 64 channels + pad
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xb = smem;                              // [3][256 rows x 128 B], XOR-swizzled 16-byte slots
    char* const wb = smem + 3 * KS_XBYTES;              // [2][8 row blocks][4 k-slices][64 lanes][16 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, n = lane & 31;
    // block -> (position tile, cout block): the cout blocks of a tile read the same input rows, so they must share an L2 -- consecutive LOGICAL
    // ids on one XCD (the bijection of conv3d_igemm_kernel; hardware block b runs on XCD b % 8), not consecutive hardware ids
    unsigned bid = blockIdx.x;
    if (p.xcd) {
        const unsigned nx = 8, q = gridDim.x / nx, r = gridDim.x % nx;
        const unsigned xcd = bid % nx, k = bid / nx;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int cb = bid % p.ncb;
    const unsigned pos0 = (bid / p.ncb) * 256u;
    const int wq = wave & 3, ph = wave >> 2;            // cout quarter (row blocks 2 wq, 2 wq + 1) / position half (tiles 4 ph .. 4 ph + 3)
    // ---- chunk copies by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass; one wave instruction lands 64 x 16 B =
    // 8 input rows, lane-linear, so the XOR swizzle of the fragment reads is applied on the SOURCE side: lane (row, physical slot) fetches the
    // row's logical slot phys ^ ((row >> 1) & 7)).  Wave w copies rows 32 w .. 32 w + 31 (4 pieces) and 4 of the 32 1-KB weight pieces ----
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const char* xsrc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = wave * 32 + u * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        const unsigned pos = min(pos0 + (unsigned)row, p.npos - 1u);
        unsigned ip = pos;
        if (p.stride == 2) {
            const unsigned f = pos / p.how, rem = pos - f * p.how;
            const unsigned oh = __umulhi(rem, p.wo_magic), ow = rem - oh * (unsigned)p.Wo;
            ip = (f * (unsigned)p.H + 2u * oh) * (unsigned)p.W + 2u * ow;
        }
        xsrc[u] = p.x + (size_t)ip * (unsigned)p.in_cs * 2u + slot * 16;
    }
    const char* wsrc = p.w + (size_t)cb * 8 * 4096 + (size_t)(wave * 4) * 1024 + lane * 16;
    const size_t wstep = (size_t)p.mb_total * 4096;     // bytes between consecutive K chunks of the packed weights
    char* const xdma = xb + wave * (32 * 128);          // this wave's rows inside an input buffer
    char* const wdma = wb + wave * 4096;                // this wave's pieces inside a weight buffer
#define KS_DMA_X(KC_, BUF_)                                                                                  \
    {                                                                                                        \
        const size_t ko_ = (size_t)min((KC_), p.kchunks - 1) * 128;      /* (past the end: the last chunk again, into a buffer nobody reads) */ \
        _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_)                                                     \
            __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[u_] + ko_), (lptr_t)(xdma + (BUF_) * KS_XBYTES + u_ * 1024), 16, 0, 0); \
    }
#define KS_DMA_W(KC_, BUF_)                                                                                  \
    {                                                                                                        \
        const char* ws_ = wsrc + (size_t)min((KC_), p.kchunks - 1) * wstep;                                  \
        _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_)                                                     \
            __builtin_amdgcn_global_load_lds((gptr_t)(ws_ + u_ * 1024), (lptr_t)(wdma + (BUF_) * KS_WBYTES + u_ * 1024), 16, 0, 0); \
    }
#define KS_COMPUTE(XBUF_, WBUF_)                                                                             \
    {                                                                                                        \
        const char* xc_ = xb + (XBUF_) * KS_XBYTES + b_base;                                                 \
        const char* wc_ = wb + (WBUF_) * KS_WBYTES + a_off;                                                  \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) {                                                \
            uint4 a_[2], b_[4];                                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) a_[i_] = *(const uint4*)(wc_ + (i_ * 4 + ks_) * 1024); \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) b_[j_] = *(const uint4*)(xc_ + j_ * 4096 + b_slot[ks_]); \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                 \
                _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) Mma<DAT_BF16>::step(a_[i_], b_[j_], acc[i_][j_]); \
        }                                                                                                    \
    }
    KS_DMA_W(0, 0);
    KS_DMA_X(0, 0);
    KS_DMA_X(1, 1);
    f32x16_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int a_off = (2 * wq) * 4096 + lane * 16;
    // B fragment of (position tile j, k-slice ks): row ph * 128 + j * 32 + n, logical slot 2 ks + khalf; the swizzle term (row >> 1) & 7 only
    // depends on n, so the offset is  b_base + j * 4096 + b_slot[ks]
    const int b_base = (ph * 128 + n) * 128;
    int b_slot[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_slot[ks] = (((ks * 2 + khalf) ^ (n >> 1)) & 7) << 4;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // chunk 0 (weights + rows) has landed; the rows of chunk 1 may still be on their way
    __syncthreads();
    // chunk kc lives in input buffer kc % 3 and weight buffer kc & 1.  Iteration kc requests the weights of chunk kc + 1 (L2-resident: one
    // iteration ahead) and the input rows of chunk kc + 2 (HBM: two iterations ahead -- an iteration of 32 MFMAs per wave is shorter than the
    // memory latency under load), computes chunk kc, then waits until only the four row pieces it has just requested are outstanding
    // (vmcnt counts in issue order) and publishes chunk kc + 1 with the barrier.  The buffers a request overwrites were last read in
    // iteration kc - 1, behind the previous barrier.
    int xi = 0;                                         // kc % 3
    for (int kc = 0; kc < p.kchunks; ++kc) {
        const int x2 = xi >= 1 ? xi - 1 : 2;            // (kc + 2) % 3
        if (!(p.ablate & 2)) KS_DMA_W(kc + 1, (kc + 1) & 1);
        if (!(p.ablate & 1)) KS_DMA_X(kc + 2, x2);
        if (!(p.ablate & 4)) KS_COMPUTE(xi, kc & 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
        xi = xi == 2 ? 0 : xi + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // nothing may land in LDS once the epilogue reuses it
    __syncthreads();
#undef KS_DMA_X
#undef KS_DMA_W
#undef KS_COMPUTE
    // ---- epilogue ----
    char* const est = smem + wave * (32 * EPITCH);
    const int sq = lane & 7, spl = lane >> 3;           // store phase: lane = 8 channels (sq) of one of 8 positions per round
    const int c0 = cb * 256 + wq * 64 + sq * 8;         // first of this lane's 8 output channels
    float sc[8], bi[8];                                 // channels past the stored Cout: 0 / 0 (their zero weight rows give zeros)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool ok = c0 + e < p.cout;
        const int c = ok ? c0 + e : 0;
        sc[e] = ok ? (p.scale ? p.scale[c] : 1.f) : 0.f;
        bi[e] = (ok && p.bias) ? p.bias[c] : 0.f;
    }
    const unsigned cres = (unsigned)min(c0, p.out_cs - 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned tile0 = pos0 + (unsigned)(ph * 128 + j * 32);
        uint4 rr[4], ra[4];
        if (p.res_mode) {                               // residual rows of the tile requested before the transpose
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned pc = min(tile0 + (unsigned)(r * 8 + spl), p.npos - 1u);
                unsigned rpos = pc;
                if (p.res_mode == 2) {                  // nearest-2x up-sampled coarser map (FPN top-down, FPN3D.py:186-222)
                    const unsigned fr = pc / p.how, rem = pc - fr * p.how;
                    const unsigned oh = __umulhi(rem, p.wo_magic), ow = rem - oh * (unsigned)p.Wo;
                    rpos = (fr * (unsigned)(p.Ho >> 1) + (oh >> 1)) * (unsigned)(p.Wo >> 1) + (ow >> 1);
                }
                rr[r] = *(const uint4*)(p.res + ((size_t)rpos * (unsigned)p.out_cs + cres) * 2u);
                // mode 4: the addend (the other gradient contribution; when it is y itself, the row is read here and overwritten below by this same lane)
                if (p.res_mode == 4) ra[r] = *(const uint4*)(p.res2 + ((size_t)pc * (unsigned)p.out_cs + cres) * 2u);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(est + n * EPITCH + (i * 32 + g * 8 + khalf * 4) * 4) =
                    make_float4(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pl = r * 8 + spl;
            const float4 t0 = *(const float4*)(est + pl * EPITCH + sq * 32);
            const float4 t1 = *(const float4*)(est + pl * EPITCH + sq * 32 + 16);
            float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bi[e];
            if (p.res_mode == 4) {                      // mode 4: SUM with the output's present contents, MASKED by the residual operand
                const uint32_t ru[4] = {rr[r].x, rr[r].y, rr[r].z, rr[r].w}, au[4] = {ra[r].x, ra[r].y, ra[r].z, ra[r].w};
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    v[2 * e2] = res_combine4(v[2 * e2], bf2f((uint16_t)(au[e2] & 0xffff)), bf2f((uint16_t)(ru[e2] & 0xffff)));
                    v[2 * e2 + 1] = res_combine4(v[2 * e2 + 1], bf2f((uint16_t)(au[e2] >> 16)), bf2f((uint16_t)(ru[e2] >> 16)));
                }
            } else if (p.res_mode) {                    // modes 1 / 2: Sum; mode 3: MASK by the forward input of the conv whose data gradient this is
                const uint32_t ru[4] = {rr[r].x, rr[r].y, rr[r].z, rr[r].w};
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    v[2 * e2] = res_combine(v[2 * e2], bf2f((uint16_t)(ru[e2] & 0xffff)), p.res_mode);
                    v[2 * e2 + 1] = res_combine(v[2 * e2 + 1], bf2f((uint16_t)(ru[e2] >> 16)), p.res_mode);
                }
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            const unsigned pos = tile0 + (unsigned)pl;
            if (pos < p.npos && c0 < p.cout)
                *(uint4*)(p.y + ((size_t)pos * (unsigned)p.out_cs + (unsigned)c0) * 2u) =
                    make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- big-tile 3x3 kernel: 256 output channels x 256 positions per block, one block per CU, one wave per SIMD ---------------------
// The 128 x 256 kernel above feeds 8 MFMAs per k-slice from 6 operand fragments per wave, two blocks per CU: the CU's operand
// delivery (L1 64 B/clk for the weight fragments, LDS for the patch) is what holds it at ~55 % of the matrix peak (DESIGN.md
// section 3).  Here a wave owns 128 channels x 128 positions -- 16 accumulators, all 256 AGPRs -- so a k-slice is 16 MFMAs
// (512 matrix cycles) for 4 + 4 fragments: half the operand traffic per MFMA, and the CU's L1 / LDS run at 50 % / 25 % of their
// rate.  With one wave per SIMD nothing hides latencies by itself, so the main loop is software-pipelined BY HAND in volatile asm
// (the compiler neither re-orders it nor inserts conservative waits):
//   * weight fragments: straight from global memory (fragment-order layout) into an R-step register ring; the slot a k-slice just
//     consumed is re-loaded with the slice R steps ahead (R x 512 cycles for L2 latency), waited for with COUNTED vmcnt;
//   * patch fragments: ds_read_b128 one step ahead (2-slot ring), counted lgkmcnt;
//   * the (kt, channel-chunk) patches are double-buffered in LDS by LDS-DMA: the next patch is requested when the current one
//     starts (36 k-slices = 18 k cycles earlier); one barrier per patch.  Every wave issues the same number of DMA pieces (the
//     tail re-fetches the last piece), so that the vmcnt arithmetic is the same in all waves.
// The launcher uses it for 256-channel-multiple layers whose grid fills the chip at least ~2 times (FPN P2 / P3 outputs).
template <int TWL, int R>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_bt_kernel(const ConvParams p) {
    constexpr int TW = 1 << TWL, TH = 256 >> TWL, PW = TW + 2, PH = TH + 2;
    constexpr int NPIX = PH * PW, NPIECE = (NPIX * 8 + 63) / 64, PBYTES = NPIECE * 1024, UMAX = (NPIECE + 3) / 4;
    static_assert(36 % R == 0 && PBYTES < 65536, "ring slots are static per unrolled step; 16-bit patch addresses");
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (p.clk) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_n = wave & 1, wave_p = wave >> 1;
    const int khalf = lane >> 5, n = lane & 31;

    // ---- XCD-aware block -> (channel block, tile, frame) map, as in conv3d_igemm_kernel ----
    unsigned bid = blockIdx.x;
    {
        const unsigned nx = 8, q = p.nblocks / nx, r = p.nblocks % nx;
        const unsigned xcd = bid % nx, k = bid / nx;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int nb = bid % p.nblk_n;
    unsigned tile = bid / p.nblk_n;
    const int fc = tile % p.otn; tile /= p.otn;
    const int tw_i = tile % p.tiles_w; tile /= p.tiles_w;
    const int th_i = tile % p.tiles_h;
    const int clip = tile / p.tiles_h;
    const int f = clip * p.otn + fc;
    const int t = p.ot0 + fc;
    const int f_in = clip * p.T + t;
    const int oh0 = th_i * TH, ow0 = tw_i * TW;

    // valid temporal taps of this output frame, rotated start (same order as the generic kernel: bit-identical sums)
    int kt_lo = 0, kt_hi = p.KT - 1;
    while (kt_lo < p.KT && (t + kt_lo - p.pt) < p.in_lo) ++kt_lo;
    while (kt_hi >= 0 && (t + kt_hi - p.pt) >= p.in_hi) --kt_hi;
    const int n_kt = kt_hi - kt_lo + 1;
    const int npat = n_kt * p.n_cchunks;
    const int kt_hi_x = kt_lo + n_kt;
    int kshift = 0;
    if (n_kt == p.KT && (DAT_KT_ROTATE)) kshift = (p.KT - (t + kt_lo - p.pt) % p.KT) % p.KT;
    int kt = kt_lo + kshift, cc = 0;
    if (kt >= kt_hi_x) kt -= n_kt;

    // ---- swizzled LDS address (k-slice 0) of the B fragment of every (tap, position sub-tile j), two per register ----
    // sub-tile j of this wave: 16 x 16 tiles: rows 8*wave_p + 2*j + (n >> 4), column n & 15; 8 x 32 tiles: row 4*wave_p + j, column n
    unsigned qp[9][2];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = TWL == 5 ? 4 * wave_p + j : 8 * wave_p + 2 * j + (n >> 4), c = TWL == 5 ? n : (n & 15);
            const int row = (r + tp / 3) * PW + c + tp % 3;
            const int g = (row >> 1) & 7;
            const unsigned a16 = (unsigned)(row * PPITCH) + (unsigned)(((khalf ^ (g & 1)) << 4) | ((g >> 1) << 5));
            if (j & 1) qp[tp][j >> 1] |= a16 << 16; else qp[tp][j >> 1] = a16;
        }
    // ---- weights: fragment (tap, chunk, 32-row block, k-slice) = 1 KiB at (((tap * ncc + chunk) * MB + block) * 4 + k-slice) * 1024 ----
    const size_t wd_cc_stride = (size_t)(p.Cout_pad >> 5) * 4096;
    const size_t tap_stride = (size_t)p.n_cchunks * wd_cc_stride;
    const char* const wd_base = p.w + (size_t)(nb * 8 + wave_n * 4) * 4096;        // this wave's four 32-row blocks
    unsigned aoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)lane * 16u + (unsigned)i * 4096u;

    // patch (KT_, CC_) of this tile -> LDS buffer B_: every wave issues exactly UMAX pieces (wave, wave + 4, ...; indices past the
    // last piece re-fetch it); lane-linear LDS-DMA image, XOR swizzle on the source side, halo / tail lanes fetch zeros
#define BT_DMA(KT_, CC_, B_)                                                                                              \
    {                                                                                                                     \
        const int fin_ = f_in + (KT_) - p.pt;                                                                             \
        const char* xf_ = p.x + ((size_t)fin_ * p.H * p.W) * p.Cin * 2 + (size_t)(CC_) * 128;                             \
        _Pragma("unroll") for (int u_ = 0; u_ < UMAX; ++u_) {                                                             \
            const int piece_ = min(wave + 4 * u_, NPIECE - 1);                                                            \
            const int it_ = piece_ * 64 + lane, row_ = it_ >> 3;                                                          \
            const int slot_ = (it_ ^ (row_ >> 1)) & 7;                                                                    \
            const int prow_ = row_ / PW, pcol_ = row_ - prow_ * PW;                                                       \
            const int ih_ = oh0 - 1 + prow_, iw_ = ow0 - 1 + pcol_;                                                       \
            const bool ok_ = row_ < NPIX && (unsigned)ih_ < (unsigned)p.H && (unsigned)iw_ < (unsigned)p.W;               \
            const char* src_ = ok_ ? xf_ + ((size_t)(unsigned)(ih_ * p.W + iw_) * (unsigned)(p.Cin * 2) + (unsigned)(slot_ * 16)) : p.zeros; \
            __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(smem + (B_) * PBYTES + piece_ * 1024), 16, 0, 0);     \
        }                                                                                                                 \
    }
#define BT_WPATCH(KT_, CC_) (wd_base + ((size_t)((KT_) * 9) * p.n_cchunks + (CC_)) * wd_cc_stride)
    // A fragments of step S_ (tap S_ / 4, k-slice S_ % 4) of the patch whose first tap sits at WP_ -> ring slot SLOT_
#define BT_ALOAD(WP_, S_, SLOT_)                                                                                          \
    {                                                                                                                     \
        const char* wt_ = (WP_) + (size_t)((S_) / 4) * tap_stride;                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                  \
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(aq[SLOT_][i_]) : "v"(aoff[i_]), "s"(wt_), "n"(((S_) % 4) * 1024) : "memory"); \
    }
    // B fragments of step S_ from buffer offset BOFF_ -> ring slot S_ % 2
#define BT_BREAD(S_, BOFF_)                                                                                               \
    {                                                                                                                     \
        if ((S_) % 4 == 0) {                                                                                              \
            unsigned q0_ = qp[(S_) / 4][0], q1_ = qp[(S_) / 4][1];                                                        \
            asm volatile("" : "+v"(q0_), "+v"(q1_));   /* opaque: keeps the unpacked / xor-ed addresses out of registers */ \
            ba[0] = (q0_ & 0xffffu) + (BOFF_); ba[1] = (q0_ >> 16) + (BOFF_);                                             \
            ba[2] = (q1_ & 0xffffu) + (BOFF_); ba[3] = (q1_ >> 16) + (BOFF_);                                             \
        }                                                                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                  \
            asm volatile("ds_read_b128 %0, %1" : "=v"(bq[(S_) % 2][j_]) : "v"(ba[j_] ^ (unsigned)(((S_) % 4) << 5)) : "memory"); \
    }
#define BT_MFMA(ACC_, A_, B_) asm volatile(DAT_MFMA16_OP " %0, %1, %2, %0" : "+a"(ACC_) : "v"(A_), "v"(B_) : "memory")

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4_t aq[R][4], bq[2][4];
    unsigned ba[4] = {0, 0, 0, 0};

    if (npat > 0) {
        // ---- prologue: first patch, first R steps of weights ----
        BT_DMA(kt, cc, 0);
        const char* wcur = BT_WPATCH(kt, cc);
        static_for(std::make_integer_sequence<int, R>{}, [&](auto ic_) __attribute__((always_inline)) {
            constexpr int s0 = decltype(ic_)::value;
            (void)&aq; (void)&aoff;                            // (operands of asm statements alone do not capture)
            BT_ALOAD(wcur, s0, s0);
        });
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * R) : "memory");      // the DMA pieces (older than the R x 4 weight loads)
        __syncthreads();
        int buf = 0;
        for (int pi = 0; pi < npat; ++pi, buf ^= 1) {
            // the patch after this one
            int ncc = cc + 1, nkt = kt;
            if (ncc == p.n_cchunks) { ncc = 0; if (++nkt == kt_hi_x) nkt = kt_lo; }
            const bool has_next = pi + 1 < npat;
            if (has_next) BT_DMA(nkt, ncc, buf ^ 1);
            const char* wnxt = has_next ? BT_WPATCH(nkt, ncc) : wcur;       // (past the last patch: harmless re-loads keep the counts fixed)
            const unsigned boff = (unsigned)buf * PBYTES;
            __builtin_amdgcn_s_setprio(1);
            BT_BREAD(0, boff);
            static_for(std::make_integer_sequence<int, 36>{}, [&](auto ic_) __attribute__((always_inline)) {
                constexpr int st = decltype(ic_)::value;
                (void)&aq; (void)&aoff; (void)&bq; (void)&ba; (void)&acc; (void)&qp;   // (asm operands alone do not capture)
                if (st + 1 < 36) BT_BREAD(st + 1, boff);
                // LDS returns in order: all but the 4 reads of the next step have landed
                if (st + 1 < 36) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // vector memory returns in order: younger than this step's weight loads are the loads of the R - 1 later steps and,
                // while those loads still date from the previous patch (st < R), this patch's UMAX DMA pieces
                if constexpr (st < R) {
                    if (has_next) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (R - 1) + UMAX) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (R - 1)) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (R - 1)) : "memory");
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) BT_MFMA(acc[i][j], aq[st % R][i], bq[st % 2][j]);
                // re-load the consumed slot with the step R ahead (this patch, or the first steps of the next one)
                if constexpr (st + R < 36) { BT_ALOAD(wcur, st + R, st % R); } else { BT_ALOAD(wnxt, st + R - 36, st % R); }
            });
            __builtin_amdgcn_s_setprio(0);
            // every wave's pieces of the next patch have landed (its vmcnt waits from step R on cover them) and it has left this buffer
            __syncthreads();
            kt = nkt; cc = ncc; wcur = wnxt;
        }
    }
    // (the asm MFMAs / loads are invisible to the compiler: drain them and cover the XDL-write -> VALU-read wait states by hand)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#undef BT_MFMA
#undef BT_BREAD
#undef BT_ALOAD
#undef BT_WPATCH
#undef BT_DMA

    // ---- epilogue: per-wave LDS transpose of 32 positions x 64 channels at a time, affine + residual + ReLU, 16-byte stores ----
    constexpr int EPITCH = 64 * 4 + 16;
    char* const est = smem + wave * (32 * EPITCH);             // (the patch buffers are free: the loop ended with a barrier)
    const int sl_c = (lane & 7) * 8, sl_p = lane >> 3;
    const size_t tile_pos = ((size_t)f * p.Ho + oh0) * p.Wo + ow0;
    char* const ybase = p.y + tile_pos * p.out_cs * 2;
    const char* const rbase = p.res_mode == 2 ? p.res + (size_t)f * (p.Ho >> 1) * (p.Wo >> 1) * p.out_cs * 2 : p.res + tile_pos * p.out_cs * 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                              // 64-channel half of this wave's 128
        const int c_st = nb * 256 + wave_n * 128 + h * 64 + sl_c;
        float sc[8], bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = (c_st + e) < p.Cout;
            sc[e] = (p.scale && ok) ? p.scale[c_st + e] : 1.f;
            bi[e] = (p.bias && ok) ? p.bias[c_st + e] : 0.f;
        }
        // residual rows one sub-tile ahead of their use
        uint4 rq[2][4];
        const bool res_on = p.res_mode && c_st < p.Cout;
        auto res_fetch = [&](int j, uint4* dst) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pl = q * 8 + sl_p;
                const int ohl = TWL == 5 ? 4 * wave_p + j : 8 * wave_p + 2 * j + (pl >> 4), owl = TWL == 5 ? pl : (pl & 15);
                const int oh = oh0 + ohl, ow = ow0 + owl;
                const bool live = oh < p.Ho && ow < p.Wo && !(p.ablate & 4);
                unsigned rpos = (unsigned)(ohl * p.Wo + owl);
                if (p.res_mode == 2) rpos = (unsigned)((oh >> 1) * (p.Wo >> 1) + (ow >> 1));
                dst[q] = *(const uint4*)(rbase + (live ? (rpos * (unsigned)p.out_cs + (unsigned)c_st) * 2u : 0u));
            }
        };
        if (res_on) res_fetch(0, rq[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (res_on && j + 1 < 4) res_fetch(j + 1, rq[(j + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4*)(est + n * EPITCH + (i * 32 + g * 8 + khalf * 4) * 4) =
                        make_float4(acc[2 * h + i][j][g * 4 + 0], acc[2 * h + i][j][g * 4 + 1], acc[2 * h + i][j][g * 4 + 2], acc[2 * h + i][j][g * 4 + 3]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pl = q * 8 + sl_p;
                const float4 t0 = *(const float4*)(est + pl * EPITCH + sl_c * 4);
                const float4 t1 = *(const float4*)(est + pl * EPITCH + sl_c * 4 + 16);
                float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                const int ohl = TWL == 5 ? 4 * wave_p + j : 8 * wave_p + 2 * j + (pl >> 4), owl = TWL == 5 ? pl : (pl & 15);
                const int oh = oh0 + ohl, ow = ow0 + owl;
                if (oh >= p.Ho || ow >= p.Wo || c_st >= p.Cout || (p.ablate & 4)) continue;
                const unsigned lpos = (unsigned)(ohl * p.Wo + owl);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bi[e];
                if (p.res_mode) {
                    const uint4 r = rq[j & 1][q];
                    const uint32_t ru[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        v[2 * e2] += bf2f((uint16_t)(ru[e2] & 0xffff));
                        v[2 * e2 + 1] += bf2f((uint16_t)(ru[e2] >> 16));
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *(uint4*)(ybase + (lpos * (unsigned)p.out_cs + (unsigned)c_st) * 2u) =
                    make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (p.clk && threadIdx.x == 0) {
        atomicAdd(&p.clk[0], __builtin_amdgcn_s_memtime() - clk_c0);
        atomicAdd(&p.clk[1], __builtin_amdgcn_s_memrealtime() - clk_r0);
    }
}

}  // namespace

namespace dat_conv {

// weights-stationary persistent kernel (conv3x3_c64_ws_kernel): what it assumes, and the tile shape with the least wasted work
bool ws64_eligible(const dat_ctx* ctx, const dat_conv_desc* d) {
    return ctx->dbg_ws64 && d->dtype == DAT_BF16 && d->Cin == 64 && d->Cout == 64 && d->KT == 1 && d->KH == 3 && d->KW == 3 &&
           d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1 && d->pad_t == 0 && d->res_mode != 2 &&
           d->out_tn <= 0 && d->out_cstride % 8 == 0 && weights_direct(ctx, d);
}

// CUs a persistent HBM-bound kernel (one block per CU) takes: all of them, or DAT_PERSIST_PCT percent (an experiment: does leaving CUs to the
// other forwards' MFMA kernels raise the step rate?), rounded down to a multiple of `mult`
int ctx_num_cu(dat_ctx* ctx);
static long long persist_cus(dat_ctx* ctx, int mult) {
    long long n = ctx_num_cu(ctx);
    if (ctx->dbg_persist_pct > 0 && ctx->dbg_persist_pct < 100) n = std::max<long long>(mult, n * ctx->dbg_persist_pct / 100 / mult * mult);
    return n;
}

int ctx_num_cu(dat_ctx* ctx) {
    if (ctx->num_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        ctx->num_cu = n > 0 ? n : 256;
    }
    return ctx->num_cu;
}

// big-tile kernel (conv3x3_bt_kernel): what it assumes; the grid must fill the chip about twice (one block per CU at a time)
bool bt_eligible(const dat_ctx* ctx, const dat_conv_desc* d) {
    return ctx->dbg_bt && d->dtype == DAT_BF16 && d->Cin % 64 == 0 && cout_pad_of(d) % 256 == 0 && d->Cout % 8 == 0 && d->KH == 3 && d->KW == 3 &&
           d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1 && d->out_cstride % 8 == 0 && weights_direct(ctx, d);
}

int bt_tile_twl(const ConvParams& p, long long* nblocks) {
    int best = 4;
    long long best_tiles = -1;
    for (int twl = 4; twl <= 5; ++twl) {
        const long long tiles = cdiv_ll(p.Ho, 256 >> twl) * cdiv_ll(p.Wo, 1 << twl);
        if (best_tiles < 0 || tiles < best_tiles) { best_tiles = tiles; best = twl; }
    }
    *nblocks = best_tiles * p.frames * (p.Cout_pad / 256);
    return best;
}

int launch_bt(dat_ctx* ctx, hipStream_t st, ConvParams& p) {
    long long nblocks = 0;
    const int twl = bt_tile_twl(p, &nblocks);
    const int tw = 1 << twl, th = 256 >> twl;
    p.th_log2 = 8 - twl; p.tw_log2 = twl; p.tile_w = tw;
    p.tiles_h = (int)cdiv_ll(p.Ho, th); p.tiles_w = (int)cdiv_ll(p.Wo, tw);
    p.n_cchunks = p.Cin / 64;
    p.nblk_n = p.Cout_pad / 256;
    p.ksplit = 1; p.part = nullptr;
    p.ablate = ctx->dbg_ablate;
    DAT_ENFORCE(ctx, nblocks > 0 && nblocks < (1ll << 31), "conv3d: grid of %lld blocks unsupported", nblocks);
    DAT_ENFORCE(ctx, (long long)p.Ho * p.Wo * std::max(p.out_cs, p.Cout) * 4 < (1ll << 31) && (long long)p.H * p.W * p.Cin * 2 < (1ll << 32),
                "conv3d: one frame of %dx%d exceeds the 32-bit offsets of the big-tile kernel", p.H, p.W);
    p.nblocks = (unsigned)nblocks;
    const int npiece = ((th + 2) * (tw + 2) * 8 + 63) / 64;
    const size_t lds = (size_t)2 * npiece * 1024;
#define BT_LAUNCH(TWL_, R_)                                                                                            \
    {                                                                                                                  \
        if (dat_ensure_lds(ctx, (const void*)conv3x3_bt_kernel<TWL_, R_>, 160 * 1024) != DAT_OK) return DAT_ERR_LAUNCH; \
        hipLaunchKernelGGL((conv3x3_bt_kernel<TWL_, R_>), dim3(p.nblocks), dim3(NTHREADS), lds, st, p);                \
    }
    // (a 9-step weight ring was tried: 256 VGPRs, spills -- and a scratch access in the loop would break the counted vmcnt waits)
    if (twl == 5) BT_LAUNCH(5, 6) else BT_LAUNCH(4, 6)
#undef BT_LAUNCH
    DAT_CHECK_LAUNCH(ctx, "conv3x3_bt");
    return DAT_OK;
}

// weights-stationary 1x1 64 -> 256 kernel (conv1x1_k64_c256_ws_kernel)
bool pw256_eligible(const dat_ctx* ctx, const dat_conv_desc* d) {
    return ctx->dbg_ws64 && d->dtype == DAT_BF16 && d->Cin == 64 && d->Cout == 256 && d->KT == 1 && d->KH == 1 && d->KW == 1 &&
           d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 0 && d->pad_w == 0 && d->pad_t == 0 && d->out_tn <= 0 &&
           d->out_cstride % 8 == 0 && weights_direct(ctx, d) &&
           // the kernel's 32-bit position / byte arithmetic and its row split by multiplication
           (long long)d->H * d->W >= 32 && (long long)d->H * d->W * d->W < (1ll << 32) &&
           (long long)d->frames * d->H * d->W * std::max(d->out_cstride, 64) * 2 < (1ll << 32);
}

int launch_pw256(dat_ctx* ctx, hipStream_t st, const ConvParams& cp) {
    Pw256Params p;
    p.x = cp.x; p.w = cp.w; p.scale = cp.scale; p.bias = cp.bias; p.res = cp.res; p.y = cp.y;
    p.npos = (long long)cp.frames * cp.H * cp.W;
    p.H = cp.H; p.W = cp.W; p.out_cs = cp.out_cs; p.relu = cp.relu; p.res_mode = cp.res_mode;
    const long long ntiles = cdiv_ll(p.npos, 32);
    DAT_ENFORCE(ctx, ntiles > 0 && ntiles < (1ll << 31), "conv3d: %lld wave tiles unsupported", ntiles);
    p.ntiles = (int)ntiles;
    p.hw = (unsigned)(cp.H * cp.W);
    p.w_magic = cp.W == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)cp.W - 1) / (unsigned)cp.W);
    const unsigned grid = (unsigned)std::min<long long>(cdiv_ll(ntiles, 4), persist_cus(ctx, 8));
    const size_t lds = (size_t)4 * 32 * (256 * 4 + 16);
    if (dat_ensure_lds(ctx, (const void*)conv1x1_k64_c256_ws_kernel, 160 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
    hipLaunchKernelGGL(conv1x1_k64_c256_ws_kernel, dim3(grid), dim3(NTHREADS), lds, st, p);
    DAT_CHECK_LAUNCH(ctx, "conv1x1_k64_c256_ws");
    return DAT_OK;
}

// weights-in-LDS 1x1 kernel (conv1x1_lw_kernel): layers whose weights fit LDS in <= 4 cout parts and that have enough positions
// for a persistent grid to amortise the weight copy (>= 2 wave tiles per wave)
static int lw_nsplit(const dat_conv_desc* d) {
    const long long wbytes = (long long)d->Cin * cout_pad_of(d) * 2;
    int ns = 1;
    while (wbytes / ns > 128 * 1024) ns *= 2;
    return ns;
}

bool pwlw_eligible(const dat_ctx* ctx, const dat_conv_desc* d) {
    if (!(ctx->dbg_pwlw && d->dtype == DAT_BF16 && d->KT == 1 && d->KH == 1 && d->KW == 1 && d->pad_h == 0 && d->pad_w == 0 && d->pad_t == 0 &&
          d->out_tn <= 0 && d->stride_h == d->stride_w && (d->stride_h == 1 || d->stride_h == 2) && weights_direct(ctx, d)))
        return false;
    const int kc = d->Cin / 64;
    if (d->Cin % 64 || !(kc == 1 || kc == 2 || kc == 4 || kc == 8)) return false;
    const int ns = lw_nsplit(d), mbt = cout_pad_of(d) / 32;
    if (ns > 4 || mbt % ns) return false;
    const int mbp = mbt / ns;
    if (!(mbp == 2 || mbp == 4 || mbp == 8 || mbp == 16) || (kc == 8 && mbp > 4)) return false;
    if ((long long)kc * mbp * 4096 + mbp * 256 + 4 * 32 * 144 > 160 * 1024) return false;
    if (d->out_cstride % 8 || d->out_cstride < d->Cout) return false;
    int Ho, Wo;
    dat_conv3d_out_shape(d, &Ho, &Wo);
    const long long npos = (long long)d->frames * Ho * Wo;
    if ((long long)Ho * Wo < 32 || (long long)Ho * Wo * Wo >= (1ll << 32)) return false;
    if (npos >= (1ll << 31) || (long long)d->frames * d->H * d->W >= (1ll << 31)) return false;
    // enough work per wave: 256 CUs x 4 waves, at least 2 tiles each per cout part
    return npos / 32 >= 2ll * 4 * ctx->num_cu / ns;
}

template <int KC, int MBW, int NPASS>
static int launch_pwlw_t(dat_ctx* ctx, hipStream_t st, const PwLwParams& p, unsigned grid, size_t lds) {
    if (p.res_mode == 4) {
        if (dat_ensure_lds(ctx, (const void*)conv1x1_lw_kernel<KC, MBW, NPASS, 1>, 160 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
        hipLaunchKernelGGL((conv1x1_lw_kernel<KC, MBW, NPASS, 1>), dim3(grid), dim3(NTHREADS), lds, st, p);
        return DAT_OK;
    }
    if (dat_ensure_lds(ctx, (const void*)conv1x1_lw_kernel<KC, MBW, NPASS>, 160 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
    hipLaunchKernelGGL((conv1x1_lw_kernel<KC, MBW, NPASS>), dim3(grid), dim3(NTHREADS), lds, st, p);
    return DAT_OK;
}

int launch_pwlw(dat_ctx* ctx, hipStream_t st, const ConvParams& cp, const dat_conv_desc* d) {
    ctx_num_cu(ctx);
    PwLwParams p;
    p.x = cp.x; p.w = cp.w; p.scale = cp.scale; p.bias = cp.bias; p.res = cp.res; p.y = cp.y; p.res2 = cp.res2;
    p.npos = (unsigned)((long long)cp.frames * cp.Ho * cp.Wo);
    p.Ho = cp.Ho; p.Wo = cp.Wo; p.H = cp.H; p.W = cp.W; p.stride = cp.sh;
    p.in_cs = cp.Cin; p.out_cs = cp.out_cs; p.cout = cp.Cout; p.relu = cp.relu; p.res_mode = cp.res_mode;
    p.ntiles = (int)cdiv_ll(p.npos, 32);
    p.how = (unsigned)(cp.Ho * cp.Wo);
    p.wo_magic = cp.Wo == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)cp.Wo - 1) / (unsigned)cp.Wo);
    p.nsplit = lw_nsplit(d); p.mb_total = cp.Cout_pad / 32;
    const int kc = cp.Cin / 64, mbp = p.mb_total / p.nsplit;
    const int mbw = kc <= 4 ? std::min(mbp, 8) : std::min(mbp, 4), npass = mbp / mbw;
    const long long per_part = std::min<long long>(cdiv_ll(p.ntiles, 4), persist_cus(ctx, 8 * p.nsplit) / p.nsplit);
    const unsigned grid = (unsigned)(per_part * p.nsplit);
    p.xcd = ctx->dbg_pw_xcd && p.nsplit > 1 && grid % (8u * (unsigned)p.nsplit) == 0;
    const size_t lds = (size_t)kc * mbp * 4096 + (size_t)mbp * 32 * 8 + 4 * 32 * (32 * 4 + 16);
    DAT_ENFORCE(ctx, lds <= 160 * 1024, "conv1x1_lw: %zu bytes of LDS", lds);
    int rc = DAT_ERR_UNSUPPORTED;
#define LW_CASE(KC_, MBW_, NP_) if (kc == KC_ && mbw == MBW_ && npass == NP_) rc = launch_pwlw_t<KC_, MBW_, NP_>(ctx, st, p, grid, lds);
    LW_CASE(1, 2, 1) LW_CASE(1, 4, 1) LW_CASE(1, 8, 1) LW_CASE(1, 8, 2)
    LW_CASE(2, 2, 1) LW_CASE(2, 4, 1) LW_CASE(2, 8, 1) LW_CASE(2, 8, 2)
    LW_CASE(4, 2, 1) LW_CASE(4, 4, 1) LW_CASE(4, 8, 1)
    LW_CASE(8, 2, 1) LW_CASE(8, 4, 1)
#undef LW_CASE
    if (rc != DAT_OK) DAT_FAIL(ctx, rc, "conv1x1_lw: no instantiation for K chunks %d x %d row blocks x %d passes", kc, mbw, npass);
    DAT_CHECK_LAUNCH(ctx, "conv1x1_lw");
    return DAT_OK;
}

// K-streaming 1x1 kernel: K >= 512 (DAT_CONV_PWKS chunks of 64; same-box A/B of the R-50 forward, round 6: off 175.6, K >= 1024 179.0, K >= 512
// 180.8 clips/s -- the K = 512 layers it takes from the weights-in-LDS kernel are the ones that needed two cout parts there), cout a multiple
// of 256 after padding, enough 256-position tiles to give most CUs a block (one block per CU at a time: all 160 KB of LDS)
bool pwks_eligible(const dat_ctx* ctx, const dat_conv_desc* d) {
    if (!(ctx->dbg_pwks > 0 && d->dtype == DAT_BF16 && d->KT == 1 && d->KH == 1 && d->KW == 1 && d->pad_h == 0 && d->pad_w == 0 && d->pad_t == 0 &&
          d->out_tn <= 0 && d->stride_h == d->stride_w && (d->stride_h == 1 || d->stride_h == 2) && weights_direct(ctx, d)))
        return false;
    if (d->Cin % 64 || d->Cin / 64 < ctx->dbg_pwks || cout_pad_of(d) % 256) return false;
    if (d->out_cstride % 8 || d->out_cstride < d->Cout || d->out_cstride < 8) return false;
    int Ho, Wo;
    dat_conv3d_out_shape(d, &Ho, &Wo);
    const long long npos = (long long)d->frames * Ho * Wo;
    if ((long long)Ho * Wo < 1 || (long long)Ho * Wo * Wo >= (1ll << 32)) return false;
    if (npos >= (1ll << 31) || (long long)d->frames * d->H * d->W >= (1ll << 31)) return false;
    const long long blocks = cdiv_ll(npos, 256) * (cout_pad_of(d) / 256);
    return npos >= 256 && blocks * 4 >= 3ll * ctx->num_cu;
}

int launch_pwks(dat_ctx* ctx, hipStream_t st, const ConvParams& cp) {
    ctx_num_cu(ctx);
    PwKsParams p;
    p.x = cp.x; p.w = cp.w; p.scale = cp.scale; p.bias = cp.bias; p.res = cp.res; p.y = cp.y; p.res2 = cp.res2;
    p.npos = (unsigned)((long long)cp.frames * cp.Ho * cp.Wo);
    p.Ho = cp.Ho; p.Wo = cp.Wo; p.H = cp.H; p.W = cp.W; p.stride = cp.sh;
    p.in_cs = cp.Cin; p.out_cs = cp.out_cs; p.cout = cp.Cout; p.relu = cp.relu; p.res_mode = cp.res_mode;
    p.ncb = cp.Cout_pad / 256; p.kchunks = cp.Cin / 64; p.mb_total = cp.Cout_pad / 32;
    p.how = (unsigned)(cp.Ho * cp.Wo);
    p.wo_magic = cp.Wo == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)cp.Wo - 1) / (unsigned)cp.Wo);
    p.ablate = ctx->dbg_ablate;
    p.xcd = ctx->dbg_pw_xcd && p.ncb > 1;
    const long long blocks = cdiv_ll(p.npos, 256) * p.ncb;
    DAT_ENFORCE(ctx, blocks > 0 && blocks < (1ll << 31), "conv1x1_ks: grid of %lld blocks unsupported", blocks);
    const size_t lds = (size_t)3 * KS_XBYTES + 2 * KS_WBYTES;     // 160 KB: three input buffers, two weight buffers (the epilogue reuses them)
    if (dat_ensure_lds(ctx, (const void*)conv1x1_ks_kernel, 160 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
    hipLaunchKernelGGL(conv1x1_ks_kernel, dim3((unsigned)blocks), dim3(KS_THREADS), lds, st, p);
    DAT_CHECK_LAUNCH(ctx, "conv1x1_ks");
    return DAT_OK;
}

int launch_ws64(dat_ctx* ctx, hipStream_t st, const ConvParams& cp) {
    ctx_num_cu(ctx);
    Ws64Params p;
    p.x = cp.x; p.w = cp.w; p.scale = cp.scale; p.bias = cp.bias; p.res = cp.res; p.y = cp.y; p.zeros = cp.zeros;
    p.frames = cp.frames; p.H = cp.H; p.W = cp.W; p.out_cs = cp.out_cs; p.relu = cp.relu; p.res_mode = cp.res_mode;
    p.ablate = ctx->dbg_ablate;
    DAT_ENFORCE(ctx, (long long)p.H * p.W * std::max(p.out_cs, 64) * 2 < (1ll << 31), "conv3d: frame of %dx%d exceeds 32-bit offsets", p.H, p.W);
    // 8 x 32 or 16 x 16 positions per tile: fewest rounds of the persistent grid (ties: the smaller 16 x 16 patch)
    int best_twl = 4;
    long long best_rounds = -1, best_tiles = 0;
    for (int twl = 4; twl <= 5; ++twl) {
        if ((ctx->dbg_ws64 == 2 && twl != 5) || (ctx->dbg_ws64 == 3 && twl != 4)) continue;   // DEBUG: DAT_CONV_WS64=2 / 3 force 8x32 / 16x16
        const int tw = 1 << twl, th = 256 >> twl;
        const long long tiles = (long long)p.frames * cdiv_ll(p.H, th) * cdiv_ll(p.W, tw);
        const long long rounds = cdiv_ll(tiles, ctx->num_cu);
        if (best_rounds < 0 || rounds < best_rounds) { best_rounds = rounds; best_twl = twl; best_tiles = tiles; }
    }
    const int tw = 1 << best_twl, th = 256 >> best_twl;
    p.tiles_h = (int)cdiv_ll(p.H, th); p.tiles_w = (int)cdiv_ll(p.W, tw);
    DAT_ENFORCE(ctx, best_tiles > 0 && best_tiles < (1ll << 31), "conv3d: %lld tiles unsupported", best_tiles);
    p.ntiles = (int)best_tiles;
    const int npiece = ((th + 2) * (tw + 2) * 8 + 63) / 64;
    const size_t lds = (size_t)2 * npiece * 1024 + 4 * 32 * (64 * 4 + 16);
    const unsigned grid = (unsigned)std::min<long long>(best_tiles, persist_cus(ctx, 8));
    if (best_twl == 5) {
        if (dat_ensure_lds(ctx, (const void*)conv3x3_c64_ws_kernel<5>, 160 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
        hipLaunchKernelGGL(conv3x3_c64_ws_kernel<5>, dim3(grid), dim3(NTHREADS), lds, st, p);
    } else {
        if (dat_ensure_lds(ctx, (const void*)conv3x3_c64_ws_kernel<4>, 160 * 1024) != DAT_OK) return DAT_ERR_LAUNCH;
        hipLaunchKernelGGL(conv3x3_c64_ws_kernel<4>, dim3(grid), dim3(NTHREADS), lds, st, p);
    }
    DAT_CHECK_LAUNCH(ctx, "conv3x3_c64_ws");
    return DAT_OK;
}

}  // namespace dat_conv
