// Fused 3D convolution as an MFMA implicit GEMM for gfx950 (CDNA4).
//
//   y[f, oh, ow, co] = act( sum_{kt,kh,kw,ci} x[f+kt-pt, oh*sh+kh-ph, ow*sw+kw-pw, ci] * w[kt,kh,kw][co][ci]
//                           * scale[co] + bias[co] + residual )
//
// Replaces the reference's ConvNd -> AffineChannelNd -> (Sum) -> Relu op chains
// (lib/modeling/ResNet3D.py:21-154, FPN3D.py:109-222, detector.py:410-438; elementwise semantics of
// lib/ops/affine_channel_nd_op.cu:20-32) with ONE kernel: the affine/bias, the residual Sum (also the FPN
// nearest-2x top-down Sum) and the ReLU live in the epilogue, so a conv output crosses HBM once.
//
// Design (MI355X-first, not a cuDNN translation):
//   * activations NDHWC, so the GEMM K axis (input channels of one tap) is contiguous: every LDS row is
//     one 128-byte line = 64 bf16 / 32 fp32 channels of one input position;
//   * the block stages an INPUT PATCH (output tile + halo) in LDS once per (kt, channel chunk) and reuses it
//     for all KH*KW spatial taps — the im2col matrix never exists, and global->LDS traffic for activations
//     drops by ~KH*KW versus a tap-by-tap implicit GEMM; only the weight tile streams per tap
//     (double-buffered, prefetched into registers one tap ahead);
//   * MFMA operands: A = weights (rows = output channels), B = activations (cols = output positions), both
//     read from LDS as one ds_read_b128 per lane per k-slice with an XOR swizzle ((row>>1)&7) that makes
//     the 16-lane read groups conflict-free for consecutive rows;
//   * bf16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate); fp32 parity mode: v_mfma_f32_32x32x2_f32 (exact
//     fp32 fma chain) consuming the same 16-byte LDS reads as 4 k-steps;
//   * D layout (col = lane&31 = position, rows = 4 consecutive channels per register quad) gives 8/16-byte
//     channel-contiguous NDHWC stores in the epilogue;
//   * blockIdx -> tile mapping is XCD-aware: output-channel blocks of one spatial tile and neighbouring
//     tiles land on the same XCD (shared L2 for patch + weights).
#include "conv_internal.h"

using namespace dat_conv;

namespace {

// BN = output channels per block, BP = output positions per block, WAVES_N x WAVES_P = 4 waves.
// TPS = spatial taps per step: a step costs ~1 us of fixed work (barrier, weight-DMA issue, table look-ups, LDS read
// latency) whatever the tile, so thin layers (64 -> 64 channels: 8 MFMAs per wave and tap) run 3 taps per step from a
// 3 x 8 KB weight stage; wide tiles keep 1 tap per step (their weight stages would not leave room for 2 blocks per CU).
//
// WD = 1 ("weights direct", the BN = 128 variants): the weight operand never touches LDS.  The packed weights are stored in
// MFMA A-fragment order -- per (tap, 64-channel chunk, 32-row block) one 4-KiB image [k-slice][lane][16 B] -- so a wave fetches
// the fragment of (row block, k-slice) with ONE fully coalesced 1-KiB global_load_dwordx4, straight into the registers the
// MFMA reads.  Each fragment register is re-loaded for the NEXT tap right after the MFMAs that consumed it, so the loads have a
// whole tap (~1000 cycles) to land.  What this removes from a tap step: the 4 LDS-DMA pieces per wave of the weight tile
// (their ISSUE was the largest non-MFMA cost of a step), the 8 ds_read_b128 of the A fragments, the
// 32 KB weight stage and -- because nothing a tap needs is written by another wave any more -- the per-tap barrier: the block
// synchronises only around a patch reload (every KH*KW taps).
//
// NTAP > 0 (with WD): the spatial taps of a patch are a compile-time unrolled sequence (9 = dense 3x3 stride 1, 1 = 1x1), which
// takes everything tap-dependent out of the hot loop: the swizzled LDS address of every (tap, position sub-tile) B fragment is
// computed ONCE per block into NTAP x PT registers (one v_xor per ds_read remains), the tap tables are never read again (the
// s_load + s_waitcnt lgkmcnt(0) they cost per tap also drained the LDS queue), the weight pointer advances by a scalar add, and the
// per-lane source offsets of the patch LDS-DMA are kept in registers across reloads.  NTAP = 0: the generic table-driven loop
// (strided 3x3 convs with their stride-parity planes, other kernel shapes).
// (the unrolled 128-position variants are held to 168 registers -- three blocks per CU -- the 256-position ones to 256)
template <int BP, int NTAP> struct MinWaves { static constexpr int value = BP == 320 ? 1 : (NTAP > 0 && NTAP != 10 && BP == 128) ? 3 : 2; };
//
// ODT = element type of the OUTPUT and of the residual (default: the operand type DT).  ODT = fp32 with DT = bf16 is the "bf16x3"
// arithmetic mode (dat_conv_desc.dtype DAT_BF16X3): activations live in HBM as fp32, the conv reads a hi / lo bf16 SPLIT of its
// input (x = hi + lo to 2^-17, dat_split_bf16x2: per 64-channel chunk q the pixel's line 2q holds hi, line 2q + 1 lo) and hi / lo
// split weights packed as K' = 3 chunks per logical chunk -- [W_hi | W_lo | W_hi] against the input chunks [hi | hi | lo] -- so that
// the MFMA loop, untouched, accumulates x_hi*W_hi + x_hi*W_lo + x_lo*W_hi in fp32: the fp32 product to ~2^-16 relative at three
// bf16 MFMAs per k-slice instead of sixteen quarter-rate v_mfma_f32_32x32x2_f32.  Only the chunk -> source-line map (p.x3) and the
// epilogue's element type know about it.
template <int DT, int BN, int BP, int WAVES_N, int TPS = 1, int WD = 0, int NTAP = 0, int ODT = DT>
__global__ __launch_bounds__(NTHREADS, (MinWaves<BP, NTAP>::value)) void conv3d_igemm_kernel(const ConvParams p) {
    constexpr int ES = ElemOf<DT>::size;
    constexpr int OES = ElemOf<ODT>::size;
    constexpr int CK = Mma<DT>::CK;
    constexpr int WAVES_P = 4 / WAVES_N;
    constexpr int WN = BN / WAVES_N;      // channels per wave
    constexpr int WP = BP / WAVES_P;      // positions per wave
    constexpr int MT = WN / 32;
    constexpr int PT = WP / 32;
    constexpr int W_ITEMS = BN * 8 / NTHREADS;  // 16-B items of the weight tile per thread
    static_assert(MT >= 1 && PT >= 1 && W_ITEMS >= 1, "tile too small");

#ifdef DAT_CONV_TRACE
    const unsigned long long tr_k0 = __builtin_amdgcn_s_memtime();
    unsigned long long tr_k1 = tr_k0;
#endif
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (p.clk) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wbuf = smem;                                       // 2 stages x TPS taps x BN x 128 B (WD: no weight stage)
    char* patch = smem + (WD ? 0 : 2 * TPS * BN * ROWB);     // PH*PW x 128 B

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform in an SGPR: LDS-DMA bases (M0) need no per-issue v_readfirstlane
    const int wave_n = wave % WAVES_N;
    const int wave_p = wave / WAVES_N;

    // ---- XCD-aware block -> (channel block, tile) map (bijective for any grid size) ----
    unsigned bid = blockIdx.x;
    if (!(p.ablate & 16)) {
        const unsigned nx = 8, q = p.nblocks / nx, r = p.nblocks % nx;
        const unsigned xcd = bid % nx, k = bid / nx;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    int split, nb;
    unsigned tile;
    if (p.order) {   // weight-stationary order: all tiles of one (cout block, split) before the next one
        tile = bid % p.ntiles;
        bid /= p.ntiles;
        split = bid % p.ksplit;
        nb = bid / p.ksplit;
    } else {
        split = bid % p.ksplit;
        bid /= p.ksplit;
        nb = bid % p.nblk_n;
        tile = bid / p.nblk_n;
    }
    // frame index fastest: the blocks that share an input frame through the temporal taps (outputs t-1, t, t+1 of one
    // spatial tile) are neighbours in the XCD's queue, so the re-reads hit that XCD's L2 instead of HBM / MALL
    const int fc = tile % p.otn;
    tile /= p.otn;
    const int tw_i = tile % p.tiles_w;
    tile /= p.tiles_w;
    const int th_i = tile % p.tiles_h;
    const int clip = tile / p.tiles_h;
    const int f = clip * p.otn + fc;         // output frame (n*otn + t - ot0)
    const int t = p.ot0 + fc;
    const int f_in = clip * p.T + t;         // input frame aligned with this output frame
    const int n0 = nb * BN;
    const int TW = 1 << p.tw_log2;
    const int oh0 = th_i << p.th_log2;
    const int ow0 = tw_i * p.tile_w;         // (1 << tw_log2, except for the 320-position linear tiles)
    // input coordinate of patch cell (0,0)
    const int ih0 = oh0 * p.sh - p.ph;
    const int iw0 = ow0 * p.sw - p.pw;

    // per-lane patch row of each position sub-tile (tap offset added per tap)
    int rowbase[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int pos = wave_p * WP + j * 32 + (lane & 31);
        const int ohl = pos >> p.tw_log2, owl = pos & (TW - 1);
        rowbase[j] = ohl * p.rsh * p.PW + owl * p.rsw;
    }
    const int khalf = lane >> 5;

    f32x16_t acc[MT][PT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // valid temporal taps for this output frame
    int kt_lo = 0, kt_hi = p.KT - 1;
    while (kt_lo < p.KT && (t + kt_lo - p.pt) < p.in_lo) ++kt_lo;
    while (kt_hi >= 0 && (t + kt_hi - p.pt) >= p.in_hi) --kt_hi;
    const int n_kt = kt_hi - kt_lo + 1;
    const int ntap = p.KH * p.KW;      // weight taps per kt
    const int ntab = p.tab_n;          // == ntap, in plane order
    const int npatch = n_kt * p.n_cchunks;                 // (kt, channel chunk) patches of this output frame
    const int pi_lo = (npatch * split) / p.ksplit, pi_hi = (npatch * (split + 1)) / p.ksplit;
    const int total = (pi_hi - pi_lo) * (ntab / TPS);     // steps of TPS taps (the launcher guarantees ntab % TPS == 0)

    const size_t w_tap_stride = (size_t)p.Cout_pad * p.Cin * ES;  // bytes between taps
    const int npatch_items = p.PH * p.PW * 8;

    // ---- weight tile: global -> LDS by LDS-DMA (global_load_lds_dwordx4), no staging registers, no ds_write pass ----
    // One wave-instruction lands 64 x 16 B = 8 tile rows, lane-linear (dest = uniform base + lane*16).  The XOR swizzle
    // the fragment reads use is therefore applied on the SOURCE side: lane (row, phys slot) fetches the row's logical
    // slot phys ^ ((row >> 1) & 7) -- the same involution as swz() (cdna_hip_programming.md rule 21).
    // Item i of this thread: row (tid >> 3) + 32*i, phys slot tid & 7; (row >> 1) & 7 == (tid >> 4) & 7 for every i.
    static_assert(W_ITEMS == 2 || W_ITEMS == 4, "weight tile = 2 or 4 16-byte items per thread");
    const unsigned w_thr_off = (unsigned)(tid >> 3) * (unsigned)(p.Cin * ES) + (unsigned)(((tid & 7) ^ ((tid >> 4) & 7)) * 16);
    const unsigned w_item_stride = 32u * (unsigned)(p.Cin * ES);
    const size_t w_blk_off = (size_t)n0 * p.Cin * ES;                                             // uniform part
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define W_DMA(SRC_, DST_) __builtin_amdgcn_global_load_lds((gptr_t)(SRC_), (lptr_t)(DST_), 16, 0, 0)
#ifndef DAT_PATCH_AUX
#define DAT_PATCH_AUX 0
#endif
#define P_DMA(SRC_, DST_) __builtin_amdgcn_global_load_lds((gptr_t)(SRC_), (lptr_t)(DST_), 16, 0, DAT_PATCH_AUX)
#define W_PREFETCH(KT_, CC_, TI_, BUF_)                                                                         \
    _Pragma("unroll") for (int u_ = 0; u_ < TPS; ++u_) {                                                        \
        const char* wbase_ = p.w + ((size_t)((KT_) * ntap + p.tab_tap[(TI_) + u_]) * w_tap_stride + (size_t)(CC_) * CK * ES + w_blk_off); \
        char* wdst_ = wbuf + ((BUF_) * TPS + u_) * BN * ROWB + wave * (8 * ROWB);                               \
        W_DMA(wbase_ + w_thr_off, wdst_);                                                                       \
        W_DMA(wbase_ + (w_thr_off + w_item_stride), wdst_ + 32 * ROWB);                                         \
        if (W_ITEMS == 4) {                                                                                     \
            W_DMA(wbase_ + (w_thr_off + 2 * w_item_stride), wdst_ + 64 * ROWB);                                 \
            W_DMA(wbase_ + (w_thr_off + 3 * w_item_stride), wdst_ + 96 * ROWB);                                 \
        }                                                                                                       \
    }
    // the tile of this step was requested one step ago: retire this wave's DMAs; the barrier that follows publishes
    // every wave's part (a ds_read is ordered behind an LDS-DMA only by the issuer's vmcnt + a barrier)
#define W_COMMIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

    // per-lane LDS byte offsets of the weight fragments (lane-constant): row -> 4 k-slices, XOR-swizzled
    int a_off[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a_off[i][ks] = swz(wave_n * WN + i * 32 + (lane & 31), ks * 2 + khalf);
    // WD: this wave's A fragments in global memory: 4 KiB per (tap, channel chunk, 32-row block), this lane's 16 B inside it
    static_assert(!WD || TPS == 1, "direct weights: one tap per step");
    const size_t wd_cc_stride = (size_t)(p.Cout_pad >> 5) * 4096;
    const char* const wd_lane = p.w + (size_t)((n0 + wave_n * WN) >> 5) * 4096 + lane * 16;
#define WD_PTR(KT_, CC_, TI_) (wd_lane + ((size_t)((KT_) * ntap + p.tab_tap[(TI_)]) * p.n_cchunks + (CC_)) * wd_cc_stride)
    if constexpr (WD && NTAP > 0) {
      constexpr int NT = NTAP == 10 ? 9 : NTAP;   // NTAP = 10: the 9 taps of a 3x3 stride-2 conv on ONE dense (2*TH+1) x (2*TW+1) patch
      if (total > 0) {
        static_assert(TPS == 1, "unrolled taps: one tap per step");
        constexpr int MAXCH = patch_pieces_per_wave(NTAP, BP);   // 1-KiB patch pieces per wave (the launcher checks nchunks <= 4 * MAXCH)
        const int kt_hi_x = kt_lo + n_kt;
        int kshift = 0;
        if (n_kt == p.KT && (DAT_KT_ROTATE)) kshift = (p.KT - (t + kt_lo - p.pt) % p.KT) % p.KT;
        int kt = kt_lo + pi_lo / p.n_cchunks + kshift, cc = pi_lo % p.n_cchunks;
        if (kt >= kt_hi_x) kt -= n_kt;
        // ---- per-lane LDS address of the B fragment (k-slice 0) of every (tap, position sub-tile); k-slice ks is `^ (ks << 5)`:
        // row * 128 + ((khalf ^ (g & 1)) << 4) + (((g >> 1) ^ ks) << 5), g = (row >> 1) & 7 -- bits 5-6 of the first two terms are 0
        // (a patch is < 64 KiB: two 16-bit addresses per register -- NTAP x PT / 2 registers instead of NTAP x PT)
        constexpr bool PACK16 = NTAP != 10;                    // (the dense stride-2 patch is 70 KiB: one address per register there)
        constexpr int NQ = PACK16 ? (PT + 1) / 2 : PT;
        unsigned qp[NT][NQ];
#pragma unroll
        for (int tp = 0; tp < NT; ++tp)
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                int row = rowbase[j] + p.tab_rowoff[tp];
                if (p.lin_w > 0) {   // linear tiling: a tap that leaves the lane's map (row / column / map border) reads the zero row
                    const int gpos = ow0 + wave_p * WP + j * 32 + (lane & 31);
                    const int rem = gpos % (p.lin_h * p.lin_w);
                    const int y = rem / p.lin_w, x = rem - y * p.lin_w;
                    const int yy = y + tp / p.KW - p.KH / 2, xx = x + tp % p.KW - p.KW / 2;
                    if ((unsigned)yy >= (unsigned)p.lin_h || (unsigned)xx >= (unsigned)p.lin_w) row = p.lin_zero_row;
                }
                const int g = (row >> 1) & 7;
                const unsigned a16 = (unsigned)(row * PPITCH) + (unsigned)(((khalf ^ (g & 1)) << 4) | ((g >> 1) << 5));
                if (!PACK16) qp[tp][j] = a16;
                else if (j & 1) qp[tp][j >> 1] |= a16 << 16;
                else qp[tp][j >> 1] = a16;
            }
        // ---- per-lane source offsets of this wave's patch pieces (relative to the frame/chunk base; -1 = halo outside the frame)
        const int nchunks = (npatch_items + 63) >> 6;
        int poff[MAXCH];
#pragma unroll
        for (int u = 0; u < MAXCH; ++u) {
            const int c = wave + 4 * u;
            const int it = c * 64 + lane;
            int off = -2;                                      // -2: no such piece / item
            if (c < nchunks && it < npatch_items) {
                const int row = it >> 3;
                const int slot = (it ^ (row >> 1)) & 7;
                const int prow = p.pw_magic ? (int)__umulhi((unsigned)row, p.pw_magic) : row, pcol = row - prow * p.PW;
                const int ih = ih0 + p.tab_dy[0] + prow * p.psh, iw = iw0 + p.tab_dx[0] + pcol * p.psw;
                off = (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && row != p.lin_zero_row) ? (int)((unsigned)(ih * p.W + iw) * (unsigned)(p.Cin * ES) + (unsigned)(slot * 16)) : -1;
            }
            poff[u] = off;
        }
        // ---- weights: this wave's fragment stream; tap tp of patch (kt, cc) sits tp * tap_stride bytes behind the patch's first tap
        // (block-uniform 64-bit base in SGPRs + one 32-bit lane offset: the loads take the "SGPR base + VGPR offset + immediate" form,
        //  no per-lane 64-bit pointer arithmetic and no pointer registers per tap)
        const size_t tap_stride = (size_t)p.n_cchunks * wd_cc_stride;
        const char* const wd_base = p.w + (size_t)((n0 + wave_n * WN) >> 5) * 4096;       // uniform
        const unsigned lane16 = (unsigned)lane * 16u;
        const char* wcur = wd_base + ((size_t)(kt * ntap) * p.n_cchunks + cc) * wd_cc_stride;
        uint4 wa[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wa[i][ks] = *(const uint4*)(wcur + i * 4096 + ks * 1024 + lane16);
        const int npat = pi_hi - pi_lo;
        for (int pi = 0; pi < npat; ++pi) {
            // (bf16x3: chunk 3q + 1 reads the SAME source line as chunk 3q -- x_hi against W_lo after x_hi against W_hi -- so the patch
            //  that is in LDS is the one it needs: no reload, no barriers)
            if (!((p.ablate & 1) && pi > 0) && !(p.x3 && pi > 0 && cc % 3 == 1)) {
                __syncthreads();                               // all waves finished reading the previous patch
                const int fin = f_in + kt - p.pt;
                const char* xbase = p.x + ((size_t)fin * p.H * p.W) * p.Cin * ES + (size_t)(p.x3 ? (cc / 3) * 2 + (cc % 3 == 2) : cc) * CK * ES;
#pragma unroll
                for (int u = 0; u < MAXCH; ++u) {
                    if (poff[u] != -2) {
                        const char* src = poff[u] >= 0 ? xbase + (unsigned)poff[u] : p.zeros;
                        P_DMA(src, patch + (wave + 4 * u) * 1024);
                    }
                }
                W_COMMIT();
                __syncthreads();
            }
            // the patch after this one (its first tap's weights are fetched during this patch's last tap)
            int ncc = cc + 1, nkt = kt;
            if (ncc == p.n_cchunks) { ncc = 0; if (++nkt == kt_hi_x) nkt = kt_lo; }
            const char* wnextpatch = (pi + 1 < npat) ? wd_base + ((size_t)(nkt * ntap) * p.n_cchunks + ncc) * wd_cc_stride : wcur;
            __builtin_amdgcn_s_setprio(1);
            // (the empty asm makes the packed addresses opaque per tap: without it the compiler hoists all NTAP x PT x 4 unpacked and
            //  xor-ed addresses out of the patch loop as loop invariants -- 144 registers, spilled to scratch)
            uint4 b[2][PT];
            unsigned qa[PT];
#pragma unroll
            for (int jj = 0; jj < NQ; ++jj) asm volatile("" : "+v"(qp[0][jj]));
#pragma unroll
            for (int j = 0; j < PT; ++j) qa[j] = !PACK16 ? qp[0][j % NQ] : (j & 1) ? (qp[0][(j >> 1) % NQ] >> 16) : (qp[0][(j >> 1) % NQ] & 0xffffu);
#pragma unroll
            for (int j = 0; j < PT; ++j) b[0][j] = *(const uint4*)(patch + qa[j]);
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) {
                const char* wnext = (tp + 1 < NT) ? wcur + (size_t)(tp + 1) * tap_stride : wnextpatch;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int cur = ks & 1, nxt = cur ^ 1;
                    if (ks < 3) {
#pragma unroll
                        for (int j = 0; j < PT; ++j) b[nxt][j] = *(const uint4*)(patch + (qa[j] ^ (unsigned)((ks + 1) << 5)));
                    } else if (tp + 1 < NT) {
                        // the first fragments of the NEXT tap, behind this tap's last k-slice: no tap opens with an exposed LDS round trip
#pragma unroll
                        for (int jj = 0; jj < NQ; ++jj) asm volatile("" : "+v"(qp[tp + 1 < NT ? tp + 1 : tp][jj]));
#pragma unroll
                        for (int j = 0; j < PT; ++j) {
                            const unsigned w2 = qp[tp + 1 < NT ? tp + 1 : tp][(PACK16 ? j >> 1 : j) % NQ];
                            qa[j] = !PACK16 ? w2 : (j & 1) ? (w2 >> 16) : (w2 & 0xffffu);
                        }
#pragma unroll
                        for (int j = 0; j < PT; ++j) b[nxt][j] = *(const uint4*)(patch + qa[j]);
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < PT; ++j) Mma<DT>::step(wa[i][ks], b[cur][j], acc[i][j]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) wa[i][ks] = *(const uint4*)(wnext + i * 4096 + ks * 1024 + lane16);   // a whole tap to land
                    // keep this k-slice's reloads HERE: left alone, the scheduler sinks all eight loads of a tap behind its last MFMA and
                    // the next tap's first MFMA then waits out a full L2 round trip
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            wcur = wnextpatch;
            cc = ncc; kt = nkt;
        }
      }
    } else
    if (total > 0) {
        // temporal taps are visited in order of the INPUT frame index mod KT, not of kt: the blocks of output frames
        // t-1, t, t+1 (queue neighbours on one XCD) then stage the same input frame during the same third of their
        // lifetime, so the temporal re-reads hit that XCD's L2 instead of going back to the fabric
        const int kt_hi_x = kt_lo + n_kt;      // one past the last valid kt
        int kshift = 0;
        if (n_kt == p.KT && (DAT_KT_ROTATE)) kshift = (p.KT - (t + kt_lo - p.pt) % p.KT) % p.KT;
        int kt = kt_lo + pi_lo / p.n_cchunks + kshift, cc = pi_lo % p.n_cchunks, ti = 0;
        if (kt >= kt_hi_x) kt -= n_kt;
        uint4 wa[MT][4];      // WD: the A fragments of the current tap (k-slice ks re-loaded for the next tap after its MFMAs)
        if (WD) {
            const char* w0 = WD_PTR(kt, cc, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wa[i][ks] = *(const uint4*)(w0 + i * 4096 + ks * 1024);
        } else {
            W_PREFETCH(kt, cc, 0, 0);
        }
        const int nchunks = (npatch_items + 63) >> 6;    // 1-KiB LDS-DMA pieces (8 patch rows each)
#ifdef DAT_CONV_TRACE
        unsigned long long tr_reload = 0, tr_wait = 0, tr_bar = 0, tr_issue = 0, tr_mma = 0, tr_t;
#define TR_NOW() __builtin_readcyclecounter()
#define TR_ADD(ACC_) { const unsigned long long n_ = TR_NOW(); ACC_ += n_ - tr_t; tr_t = n_; }
        tr_t = TR_NOW();
        const unsigned long long tr_c0 = __builtin_amdgcn_s_memtime(), tr_r0 = __builtin_amdgcn_s_memrealtime();
#else
#define TR_ADD(ACC_)
#endif
        for (int step = 0; step < total; ++step) {
            // (bf16x3, one stride plane: chunk 3q + 1 re-uses the patch of chunk 3q, see the unrolled variant)
            const bool x3_same = p.x3 && p.tab_new == 1u && step > 0 && cc % 3 == 1;
            if (((p.tab_new >> ti) & 1u) && !((p.ablate & 1) && step > 0) && !(p.ablate & 8) && !x3_same) {
                __syncthreads();  // all waves finished reading the previous patch
                // ---- stage the input patch (tile + halo) of (kt, cc, plane): global -> LDS by LDS-DMA, every piece of
                // the patch in flight at once (no staging registers, no ds_write pass).  Lane (row, phys slot) fetches the
                // pixel's logical 16-B slot phys ^ ((row >> 1) & 7); halo pixels outside the frame fetch zeros.
                const int fin = f_in + kt - p.pt;
                const char* xbase = p.x + ((size_t)fin * p.H * p.W) * p.Cin * ES + (size_t)(p.x3 ? (cc / 3) * 2 + (cc % 3 == 2) : cc) * CK * ES;
                const int py0 = ih0 + p.tab_dy[ti], px0 = iw0 + p.tab_dx[ti];
#pragma unroll 2
                for (int c = wave; c < nchunks; c += 4) {
                    const int it = c * 64 + lane;
                    if (it < npatch_items) {
                        const int row = it >> 3;
                        const int slot = (it ^ (row >> 1)) & 7;
                        const int prow = p.pw_magic ? (int)__umulhi((unsigned)row, p.pw_magic) : row, pcol = row - prow * p.PW;
                        const int ih = py0 + prow * p.psh, iw = px0 + pcol * p.psw;
                        const char* src = p.zeros;
                        if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                            src = xbase + ((unsigned)(ih * p.W + iw) * (unsigned)(p.Cin * ES) + (unsigned)(slot * 16));
                        P_DMA(src, patch + c * 1024);
                    }
                }
            }
            TR_ADD(tr_reload);
            if (!WD || ((p.tab_new >> ti) & 1u)) {   // WD: only a patch reload needs the block to meet
                W_COMMIT();
                TR_ADD(tr_wait);
                __syncthreads();
            }
            TR_ADD(tr_bar);
            // advance to the next (kt, cc, table entry) and prefetch its weight tile (lands during this step's MFMAs)
            int nti = ti + TPS, ncc = cc, nkt = kt;
            if (nti == ntab) {
                nti = 0;
                if (++ncc == p.n_cchunks) { ncc = 0; if (++nkt == kt_hi_x) nkt = kt_lo; }
            }
            // (issuing the pieces between the k-slices' MFMAs instead measured 7 % slower: a DMA issue stalls the MFMA stream)
            if (!WD && step + 1 < total && !((p.ablate & 2) && step > 1)) W_PREFETCH(nkt, ncc, nti, (step + 1) & 1);
            const char* const wnext = (step + 1 < total) ? WD_PTR(nkt, ncc, nti) : WD_PTR(kt, cc, ti);

            TR_ADD(tr_issue);
            // ---- compute the step's taps: 4 k-slices of 16 B per row each, fragments double-buffered in registers ----
            // the co-resident block's wave on this SIMD is usually in its load phase: let the MFMA stream win arbitration
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int u = 0; u < TPS; ++u) {
                const int tapoff = p.tab_rowoff[ti + u];
                const char* wb = wbuf + ((step & 1) * TPS + u) * BN * ROWB;
                // patch fragment of (row, k-slice ks, k-half): 16-B slot (2*ks + khalf) ^ ((row >> 1) & 7) of the row's line
                const char* bp[PT];
                int bx[PT];
#pragma unroll
                for (int j = 0; j < PT; ++j) {
                    const int row = rowbase[j] + tapoff;
                    const int g = (row >> 1) & 7;
                    bp[j] = patch + (row * PPITCH + ((khalf ^ (g & 1)) << 4));
                    bx[j] = g >> 1;
                }
                uint4 a[2][MT], b[2][PT];
                if (!WD) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) a[0][i] = *(const uint4*)(wb + a_off[i][0]);
                }
#pragma unroll
                for (int j = 0; j < PT; ++j) b[0][j] = *(const uint4*)(bp[j] + (bx[j] << 5));
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int cur = ks & 1, nxt = cur ^ 1;
                    if (ks < 3) {
                        if (!WD) {
#pragma unroll
                            for (int i = 0; i < MT; ++i) a[nxt][i] = *(const uint4*)(wb + a_off[i][ks + 1]);
                        }
#pragma unroll
                        for (int j = 0; j < PT; ++j) b[nxt][j] = *(const uint4*)(bp[j] + (((ks + 1) ^ bx[j]) << 5));
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < PT; ++j) Mma<DT>::step(WD ? wa[i][ks] : a[cur][i], b[cur][j], acc[i][j]);
                    if (WD) {   // this k-slice's fragments of the NEXT tap: a whole tap to land.  Unconditional (a branch around the loads made
                                // the compiler wait with vmcnt(0) at the top of every tap) and pinned here (see the unrolled variant)
#pragma unroll
                        for (int i = 0; i < MT; ++i) wa[i][ks] = *(const uint4*)(wnext + i * 4096 + ks * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            ti = nti; cc = ncc; kt = nkt;
            TR_ADD(tr_mma);
        }
#ifdef DAT_CONV_TRACE
        if (tid == 0 && p.dbg) {
            atomicAdd(&p.dbg[0], tr_reload); atomicAdd(&p.dbg[1], tr_wait); atomicAdd(&p.dbg[2], tr_bar);
            atomicAdd(&p.dbg[3], tr_issue); atomicAdd(&p.dbg[4], tr_mma); atomicAdd(&p.dbg[5], (unsigned long long)total);
            atomicAdd(&p.dbg[6], __builtin_amdgcn_s_memtime() - tr_c0);
            atomicAdd(&p.dbg[7], __builtin_amdgcn_s_memrealtime() - tr_r0);
        }
        tr_k1 = __builtin_amdgcn_s_memtime();
#endif
    }
#undef WD_PTR
#undef W_PREFETCH
#undef W_COMMIT
#undef W_DMA
#undef P_DMA

    // ---- epilogue: affine/bias + residual + relu, staged through LDS so that HBM sees whole 128-B+ runs ----
    // The MFMA result layout gives a lane 4 channels of ONE position (register r -> channel (r&3) + 8*(r>>2) + 4*(lane>>5)),
    // i.e. 8-byte pieces 512+ bytes apart: written directly, every store instruction touched 32-64 different lines and a
    // 256x128 tile took ~47k cycles (1x1 convs spent 80 % of their time here).  Each wave now transposes 32 positions x
    // WN channels at a time through its own LDS slice (fp32, pitch WN*4+16 B: conflict-free b128 writes) and reads it
    // back position-major, 16 B of output per lane, so one store instruction covers 4-8 complete position rows.
    static_assert(WN == 64, "epilogue staging assumes 64 channels per wave");
    constexpr int EPITCH = WN * 4 + 16;
    constexpr int CPL = 16 / OES;                 // channels per lane in the store phase (8 bf16 / 4 fp32)
    constexpr int LPP = WN / CPL;                // lanes per position (8 / 16)
    constexpr int PPI = 64 / LPP;                // positions per store instruction (8 / 4)
    __syncthreads();                             // every wave is done with the weight / patch buffers
    char* est = smem + wave * (32 * EPITCH);
    const int sl_c = (lane % LPP) * CPL;         // this lane's first channel inside the wave's 64
    const int sl_p = lane / LPP;
    const int cbase = n0 + wave_n * WN;
    const int c_st = cbase + sl_c;
    const bool part_mode = p.ksplit > 1;
    float sc[CPL], bi[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        const bool ok = (c_st + e) < p.Cout;
        sc[e] = (!part_mode && p.scale && ok) ? p.scale[c_st + e] : 1.f;
        bi[e] = (!part_mode && p.bias && ok) ? p.bias[c_st + e] : 0.f;
    }
    const size_t npos_all = (size_t)p.frames * p.Ho * p.Wo;
    // addresses = block-uniform 64-bit base (SGPRs: frame, tile origin) + a 32-bit per-lane offset inside the frame (the launcher
    // checks that one frame of output / partials is < 2 GB): the stores take the "SGPR base + VGPR offset" form and the
    // epilogue's per-store 64-bit multiply-add chains (a third of its VALU work) disappear
    const size_t tile_pos = ((size_t)f * p.Ho + oh0) * p.Wo + ow0;
    char* const ybase = p.y + tile_pos * p.out_cs * OES;
    const char* const rbase = p.res_mode == 2 ? p.res + (size_t)f * (p.Ho >> 1) * (p.Wo >> 1) * p.out_cs * OES
                                              : p.res + tile_pos * p.out_cs * OES;
    const char* const abase = p.res2 + tile_pos * p.out_cs * OES;      // (mode 4 only)
    float* const pbase = part_mode ? p.part + ((size_t)split * npos_all + tile_pos) * p.Cout : nullptr;
    // Residual rows (Sum shortcuts, the top-down map of the FPN laterals) are fetched one position group AHEAD of their use: a load
    // issued where its value is added costs a full memory round trip per store group -- with a top-down map the 64 -> 256 lateral
    // took 0.188 ms (cold caches) against 0.108 ms without one, for 66 MB of extra reads.
    constexpr int NQ = 32 / PPI;
    // (mode 4 -- sum with the output's present contents, masked -- takes the load-at-use path below: it reaches this kernel only for shapes the
    //  1x1 kernels do not take, and a second prefetch ring would cost every variant of this kernel 32-64 registers)
    const bool m4 = p.res_mode == 4;
    const bool res_pre = p.res_mode && !m4 && !part_mode && (p.out_cs & 7) == 0 && c_st + CPL <= p.Cout;   // whole 16-byte pieces
    uint4 rq[2][NQ];
    auto res_fetch = [&](int j, uint4* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pos = wave_p * WP + j * 32 + q * PPI + sl_p;
            const int ohl = pos >> p.tw_log2, owl = pos & (TW - 1);
            const int oh = oh0 + ohl, ow = ow0 + owl;
            const bool live = oh < p.Ho && ow < p.Wo && !(p.ablate & 4);
            unsigned rpos = (unsigned)(ohl * p.Wo + owl);
            if (p.res_mode == 2) rpos = (unsigned)((oh >> 1) * (p.Wo >> 1) + (ow >> 1));
            dst[q] = *(const uint4*)(rbase + (live ? (rpos * (unsigned)p.out_cs + (unsigned)c_st) * (unsigned)OES : 0u));
        }
    };
    if (res_pre) res_fetch(0, rq[0]);
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        if (res_pre && j + 1 < PT) res_fetch(j + 1, rq[(j + 1) & 1]);
        // phase 1: accumulators -> LDS [position][channel] fp32
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(est + (lane & 31) * EPITCH + (i * 32 + g * 8 + khalf * 4) * 4) =
                    make_float4(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
        // phase 2: position-major read back, fused epilogue, 16-byte stores
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pl = q * PPI + sl_p;                       // position inside this 32-position group
            float v[CPL];
#pragma unroll
            for (int e4 = 0; e4 < CPL / 4; ++e4) {
                const float4 t = *(const float4*)(est + pl * EPITCH + (sl_c + e4 * 4) * 4);
                v[e4 * 4 + 0] = t.x; v[e4 * 4 + 1] = t.y; v[e4 * 4 + 2] = t.z; v[e4 * 4 + 3] = t.w;
            }
            const int pos = wave_p * WP + j * 32 + pl;
            const int ohl = pos >> p.tw_log2, owl = pos & (TW - 1);
            const int oh = oh0 + ohl, ow = ow0 + owl;
            if (oh >= p.Ho || ow >= p.Wo || c_st >= p.Cout || (p.ablate & 4)) continue;
            const unsigned lpos = (unsigned)(ohl * p.Wo + owl);   // position relative to the tile origin, same frame
            const int nch = min(CPL, p.Cout - c_st);             // multiple of 4
            if (part_mode) {   // raw fp32 partial sums; splitk_finish_kernel applies the epilogue
                float* dst = pbase + (lpos * (unsigned)p.Cout + (unsigned)c_st);
#pragma unroll
                for (int e4 = 0; e4 < CPL / 4; ++e4)
                    if (e4 * 4 < nch) *(float4*)(dst + e4 * 4) = make_float4(v[e4 * 4], v[e4 * 4 + 1], v[e4 * 4 + 2], v[e4 * 4 + 3]);
                continue;
            }
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] = v[e] * sc[e] + bi[e];
            if (res_pre) {
                const uint4 r = rq[j & 1][q];
                if (ODT == DAT_BF16) {
                    const uint32_t ru[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int e2 = 0; e2 < CPL / 2; ++e2) {
                        v[2 * e2] = res_combine(v[2 * e2], bf2f((uint16_t)(ru[e2 % 4] & 0xffff)), p.res_mode);
                        v[2 * e2 + 1] = res_combine(v[2 * e2 + 1], bf2f((uint16_t)(ru[e2 % 4] >> 16)), p.res_mode);
                    }
                } else {
                    v[0] = res_combine(v[0], __uint_as_float(r.x), p.res_mode); v[1] = res_combine(v[1], __uint_as_float(r.y), p.res_mode);
                    v[2] = res_combine(v[2], __uint_as_float(r.z), p.res_mode); v[3] = res_combine(v[3], __uint_as_float(r.w), p.res_mode);
                }
            } else if (m4) {
                const unsigned off = (lpos * (unsigned)p.out_cs + (unsigned)c_st) * (unsigned)OES;
                const char* rp = rbase + off;
                const char* ap = abase + off;
                if (ODT == DAT_BF16) {
#pragma unroll
                    for (int e4 = 0; e4 < CPL / 4; ++e4)
                        if (e4 * 4 < nch) {
                            const uint2 r = *(const uint2*)(rp + e4 * 8), o = *(const uint2*)(ap + e4 * 8);
                            v[e4 * 4 + 0] = res_combine4(v[e4 * 4 + 0], bf2f((uint16_t)(o.x & 0xffff)), bf2f((uint16_t)(r.x & 0xffff)));
                            v[e4 * 4 + 1] = res_combine4(v[e4 * 4 + 1], bf2f((uint16_t)(o.x >> 16)), bf2f((uint16_t)(r.x >> 16)));
                            v[e4 * 4 + 2] = res_combine4(v[e4 * 4 + 2], bf2f((uint16_t)(o.y & 0xffff)), bf2f((uint16_t)(r.y & 0xffff)));
                            v[e4 * 4 + 3] = res_combine4(v[e4 * 4 + 3], bf2f((uint16_t)(o.y >> 16)), bf2f((uint16_t)(r.y >> 16)));
                        }
                } else {
                    const float4 r = *(const float4*)rp, o = *(const float4*)ap;
                    v[0] = res_combine4(v[0], o.x, r.x); v[1] = res_combine4(v[1], o.y, r.y);
                    v[2] = res_combine4(v[2], o.z, r.z); v[3] = res_combine4(v[3], o.w, r.w);
                }
            } else if (p.res_mode) {
                unsigned rpos = lpos;
                if (p.res_mode == 2) rpos = (unsigned)((oh >> 1) * (p.Wo >> 1) + (ow >> 1));
                const char* rp = rbase + (rpos * (unsigned)p.out_cs + (unsigned)c_st) * (unsigned)OES;
                if (ODT == DAT_BF16) {
#pragma unroll
                    for (int e4 = 0; e4 < CPL / 4; ++e4)
                        if (e4 * 4 < nch) {
                            const uint2 r = *(const uint2*)(rp + e4 * 8);
                            v[e4 * 4 + 0] = res_combine(v[e4 * 4 + 0], bf2f((uint16_t)(r.x & 0xffff)), p.res_mode);
                            v[e4 * 4 + 1] = res_combine(v[e4 * 4 + 1], bf2f((uint16_t)(r.x >> 16)), p.res_mode);
                            v[e4 * 4 + 2] = res_combine(v[e4 * 4 + 2], bf2f((uint16_t)(r.y & 0xffff)), p.res_mode);
                            v[e4 * 4 + 3] = res_combine(v[e4 * 4 + 3], bf2f((uint16_t)(r.y >> 16)), p.res_mode);
                        }
                } else {
                    const float4 r = *(const float4*)rp;
                    v[0] = res_combine(v[0], r.x, p.res_mode); v[1] = res_combine(v[1], r.y, p.res_mode);
                    v[2] = res_combine(v[2], r.z, p.res_mode); v[3] = res_combine(v[3], r.w, p.res_mode);
                }
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            char* yp = ybase + (lpos * (unsigned)p.out_cs + (unsigned)c_st) * (unsigned)OES;
            if (ODT == DAT_BF16) {
                uint32_t o[CPL / 2];
#pragma unroll
                for (int e2 = 0; e2 < CPL / 2; ++e2) o[e2] = f2bf2(v[2 * e2], v[2 * e2 + 1]);
                if (nch == CPL && (p.out_cs & 7) == 0) {
                    *(uint4*)yp = make_uint4(o[0], o[1], o[2], o[3]);
                } else {
                    *(uint2*)yp = make_uint2(o[0], o[1]);
                    if (nch > 4) *(uint2*)(yp + 8) = make_uint2(o[2], o[3]);
                }
            } else {
                *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
                if (ODT == DAT_F32 && DT == DAT_BF16 && p.y_split) {
                    // the same values as hi / lo bf16 halves for the next bf16x3 conv (bit-identical to dat_split_bf16x2 of the stored y):
                    // 64-channel chunk q of the pixel: line 2q = hi, line 2q + 1 = lo; this lane's 4 channels = 8 bytes in each line
                    const uint32_t h0 = f2bf2(v[0], v[1]), h1 = f2bf2(v[2], v[3]);
                    const uint32_t l0 = f2bf2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
                    const uint32_t l1 = f2bf2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
                    char* sp = p.y_split + ((tile_pos + lpos) * (size_t)(2 * p.out_cs) + (size_t)((c_st >> 6) * 128 + (c_st & 63))) * 2;
                    *(uint2*)sp = make_uint2(h0, h1);
                    *(uint2*)(sp + 128) = make_uint2(l0, l1);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (p.clk && threadIdx.x == 0) {   // shader clock under this kernel's own load = 100 MHz * clk[0] / clk[1]
        atomicAdd(&p.clk[0], __builtin_amdgcn_s_memtime() - clk_c0);
        atomicAdd(&p.clk[1], __builtin_amdgcn_s_memrealtime() - clk_r0);
    }
#ifdef DAT_CONV_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && p.dbg) {
        const unsigned long long e = __builtin_amdgcn_s_memtime();
        atomicAdd(&p.dbg[8], e - tr_k0);
        atomicAdd(&p.dbg[9], e - tr_k1);
    }
#endif
}

// split-K finish: y = act(sum_s part[s] * scale + bias + residual), 4 channels per thread
template <int DT>
__global__ void splitk_finish_kernel(const float* __restrict__ part, int ksplit, size_t npos, int Cout, int out_cs,
                                     const float* __restrict__ scale, const float* __restrict__ bias,
                                     const char* __restrict__ res, int res_mode, int frames, int Ho, int Wo, int relu,
                                     char* y, char* __restrict__ y_split = nullptr, const char* res2 = nullptr) {
    const int c4 = Cout >> 2;
    const size_t total = npos * c4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4) * 4;
        const size_t opos = i / c4;
        float4 a = *(const float4*)(part + opos * Cout + c);
        for (int s = 1; s < ksplit; ++s) {
            const float4 b = *(const float4*)(part + ((size_t)s * npos + opos) * Cout + c);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float v[4] = {a.x, a.y, a.z, a.w};
        if (scale) { const float4 s4 = *(const float4*)(scale + c); v[0] *= s4.x; v[1] *= s4.y; v[2] *= s4.z; v[3] *= s4.w; }
        if (bias) { const float4 b4 = *(const float4*)(bias + c); v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
        if (res_mode) {
            size_t rpos = opos;
            if (res_mode == 2) {
                const int ow = opos % Wo;
                const size_t q = opos / Wo;
                const int oh = q % Ho;
                const size_t f = q / Ho;
                rpos = (f * (Ho >> 1) + (oh >> 1)) * (Wo >> 1) + (ow >> 1);
            }
            if (res_mode == 4) {          // sum with the addend `res2` (may be y: each element is read and written by this thread), masked by `res`
                if (DT == DAT_BF16) {
                    const uint2 r = *(const uint2*)(res + (rpos * out_cs + c) * 2), o = *(const uint2*)(res2 + (opos * out_cs + c) * 2);
                    v[0] = res_combine4(v[0], bf2f((uint16_t)(o.x & 0xffff)), bf2f((uint16_t)(r.x & 0xffff)));
                    v[1] = res_combine4(v[1], bf2f((uint16_t)(o.x >> 16)), bf2f((uint16_t)(r.x >> 16)));
                    v[2] = res_combine4(v[2], bf2f((uint16_t)(o.y & 0xffff)), bf2f((uint16_t)(r.y & 0xffff)));
                    v[3] = res_combine4(v[3], bf2f((uint16_t)(o.y >> 16)), bf2f((uint16_t)(r.y >> 16)));
                } else {
                    const float4 r = *(const float4*)(res + (rpos * out_cs + c) * 4), o = *(const float4*)(res2 + (opos * out_cs + c) * 4);
                    v[0] = res_combine4(v[0], o.x, r.x); v[1] = res_combine4(v[1], o.y, r.y);
                    v[2] = res_combine4(v[2], o.z, r.z); v[3] = res_combine4(v[3], o.w, r.w);
                }
            } else if (DT == DAT_BF16) {
                const uint2 r = *(const uint2*)(res + (rpos * out_cs + c) * 2);
                v[0] = res_combine(v[0], bf2f((uint16_t)(r.x & 0xffff)), res_mode); v[1] = res_combine(v[1], bf2f((uint16_t)(r.x >> 16)), res_mode);
                v[2] = res_combine(v[2], bf2f((uint16_t)(r.y & 0xffff)), res_mode); v[3] = res_combine(v[3], bf2f((uint16_t)(r.y >> 16)), res_mode);
            } else {
                const float4 r = *(const float4*)(res + (rpos * out_cs + c) * 4);
                v[0] = res_combine(v[0], r.x, res_mode); v[1] = res_combine(v[1], r.y, res_mode);
                v[2] = res_combine(v[2], r.z, res_mode); v[3] = res_combine(v[3], r.w, res_mode);
            }
        }
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (DT == DAT_BF16) {
            uint2 o;
            o.x = f2bf2(v[0], v[1]);
            o.y = f2bf2(v[2], v[3]);
            *(uint2*)(y + (opos * out_cs + c) * 2) = o;
        } else {
            *(float4*)(y + (opos * out_cs + c) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            if (y_split) {   // (bf16x3 mode: the hi / lo split of the same values, see the conv kernel's epilogue)
                const uint32_t h0 = f2bf2(v[0], v[1]), h1 = f2bf2(v[2], v[3]);
                const uint32_t l0 = f2bf2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
                const uint32_t l1 = f2bf2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
                char* sp = y_split + (opos * (size_t)(2 * out_cs) + (size_t)((c >> 6) * 128 + (c & 63))) * 2;
                *(uint2*)sp = make_uint2(h0, h1);
                *(uint2*)(sp + 128) = make_uint2(l0, l1);
            }
        }
    }
}

// Tiles of the LINEAR position tiling (launch_conv) of a 3x3 stride-1 same-size layer at `bpv` positions per tile, or 0 when the layer
// cannot use it: kT = 1 layers tile the whole frames x H x W sequence, layers with temporal taps one strip per frame.
static long long linear_tiles(const dat_ctx* ctx, const dat_conv_desc* d, const ConvParams& p, int bpv) {
    if (!ctx->dbg_linear || !ctx->dbg_ntap || d->KH != 3 || d->KW != 3 || d->stride_h != 1 || d->stride_w != 1 || d->res_mode == 2 ||
        p.H != p.Ho || p.W != p.Wo)
        return 0;
    if (d->KT == 1 && (d->pad_t != 0 || d->out_tn > 0)) return 0;
    if (d->KT > 1 && (ctx->dbg_linear & 8)) return 0;   // (DAT_CONV_LINEAR=9: A/B switch, no per-frame strips)
    const long long per = d->KT > 1 ? (long long)p.Ho * p.Wo : (long long)p.frames * p.Ho * p.Wo;
    const long long nr = bpv + 2 * (p.Wo + 1) + 1;
    if (((nr * 8 + 63) >> 6) > 4 * patch_pieces_per_wave(9, bpv) || nr * PPITCH >= 65536 ||
        per * std::max(p.Cin, std::max(p.out_cs, p.Cout)) * 4 >= (1ll << 31))
        return 0;
    return (d->KT > 1 ? (long long)p.frames : 1) * ((per + bpv - 1) / bpv);
}

struct TileChoice {
    int th_log2, tw_log2;
};

// pick the 2^a x 2^b tile (a+b = log2(BP)) that wastes the fewest output positions, tie -> squarer patch
TileChoice choose_tile(int Ho, int Wo, int bp_log2, int sh, int sw, int KH, int KW, int force_tw) {
    if (force_tw >= 0 && force_tw <= bp_log2) return TileChoice{bp_log2 - force_tw, force_tw};
    TileChoice best{0, bp_log2};
    double best_cost = 1e30;
    for (int a = 0; a <= bp_log2; ++a) {
        const int b = bp_log2 - a;
        const long long th = 1 << a, tw = 1 << b;
        const long long tiles = cdiv_ll(Ho, th) * cdiv_ll(Wo, tw);
        const double waste = (double)(tiles * th * tw) / ((double)Ho * Wo);
        const long long PH = th + (KH - 1) / sh;
        const long long PW = tw + (KW - 1) / sw;
        const double halo = (double)(PH * PW) / (double)(th * tw);
        // a 32-lane MFMA column block spans 32/tw tile rows; with tw < 32 and a halo the LDS rows it reads are not
        // consecutive, which costs ~2-way bank conflicts on part of every ds_read_b128 (measured)
        const double conflict = (tw < 32 && KW > 1) ? 1.06 : 1.0;
        const double cost = waste * (1.0 + 0.15 * (halo - 1.0)) * conflict;
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = TileChoice{a, b};
        }
    }
    return best;
}

template <int DT, int BN, int BP, int WAVES_N, int TPS = 1, int WD = 0, int NTAP = 0, int ODT = DT>
int launch_conv(dat_ctx* ctx, hipStream_t st, ConvParams& p, int bp_log2, int ksplit, bool linear = false) {
    TileChoice tc = choose_tile(p.Ho, p.Wo, bp_log2, p.sh, p.sw, p.KH, p.KW, ctx->dbg_tw_log2);
    p.lin_h = p.lin_w = 0;
    p.lin_zero_row = -1;
    if (linear) {
        // LINEAR position tiling of small maps (RoI heads: 100 x 14 x 14): a power-of-two 2-D tile wastes 23 % of a 16 x 16 tile on a
        // 14 x 14 map.  The maps are contiguous in memory (NDHWC, frame-major), so a tile is BP CONSECUTIVE positions of the
        // frames x H x W sequence, its patch the same range widened by W + 1 positions on either side (one contiguous DMA image), and
        // tap (kh, kw) of position i reads patch row i + kh*W + kw.  Where that neighbour is not the lane's own map any more (row, column
        // or map border) the precomputed fragment address points at an all-zero patch row instead: masking costs nothing in the loop.
        // The kernel sees the whole thing as a 1 x N strip with 1 x BP tiles (its epilogue and patch loader need nothing else).
        // Layers with temporal taps (round 3: res4 / res5 / P4 / P5 of the 3-D bodies, 24 x 42 and 48 x 84 maps -- 2-D tiles of 256
        // positions cover 66 % / 87 % of what they compute there) get one strip PER FRAME instead: a tile must not span frames, whose
        // temporal taps differ; the frame / clip bookkeeping of the kernel stays as it is.
        const bool per_frame = p.KT > 1;
        const long long total = per_frame ? (long long)p.Ho * p.Wo : (long long)p.frames * p.Ho * p.Wo;
        DAT_ENFORCE(ctx, NTAP == 9 && (per_frame || p.pt == 0) && p.sh == 1 && p.sw == 1 && p.H == p.Ho && p.W == p.Wo && p.res_mode != 2 &&
                             total * std::max(p.Cin, std::max(p.out_cs, p.Cout)) * 4 < (1ll << 31),
                    "conv3d: linear tiling on an unsupported layer");
        p.lin_h = p.H; p.lin_w = p.W;
        const int wr = p.W, nr = BP + 2 * (wr + 1) + 1;
        tc = TileChoice{0, bp_log2};
        p.H = p.Ho = 1; p.W = p.Wo = (int)total;
        if (!per_frame) { p.frames = 1; p.T = 1; p.ot0 = 0; p.otn = 1; p.in_lo = 0; p.in_hi = 1; }
        p.ph = 0; p.pw = wr + 1;
        p.lin_zero_row = nr - 1;
        p.th_log2 = 0; p.tw_log2 = bp_log2;      // (bp_log2 = 9 for the 320-position tiles: pos >> tw_log2 == 0, pos & (TW - 1) == pos)
        p.tile_w = BP;
        p.tiles_h = 1; p.tiles_w = (int)cdiv_ll(total, BP);
        p.psh = p.psw = 1; p.rsh = p.rsw = 1;
        p.PH = 1; p.PW = nr;
        p.tab_new = 1u; p.tab_n = 9;
        for (int i = 0; i < 9; ++i) { p.tab_tap[i] = i; p.tab_rowoff[i] = (i / 3) * wr + i % 3; p.tab_dy[i] = p.tab_dx[i] = 0; }
    }
    const int th = 1 << tc.th_log2, tw = 1 << tc.tw_log2;
    if (!linear) {
    p.th_log2 = tc.th_log2;
    p.tw_log2 = tc.tw_log2;
    p.tile_w = tw;
    p.tiles_h = (p.Ho + th - 1) / th;
    p.tiles_w = (p.Wo + tw - 1) / tw;
    p.psh = p.sh;
    p.psw = p.sw;
    p.rsh = p.rsw = 1;
    p.PH = th + (p.KH - 1) / p.sh;
    p.PW = tw + (p.KW - 1) / p.sw;
    }
    if (linear) {
    } else if (NTAP == 10) {   // one DENSE patch for all 9 taps of a strided 3x3: neighbouring outputs are `stride` patch cells apart
        p.psh = p.psw = 1;
        p.rsh = p.sh; p.rsw = p.sw;
        p.PH = (th - 1) * p.sh + p.KH;
        p.PW = (tw - 1) * p.sw + p.KW;
        p.tab_new = 1u;
        p.tab_n = p.KH * p.KW;
        for (int i = 0; i < p.tab_n; ++i) {
            p.tab_tap[i] = i;
            p.tab_rowoff[i] = (i / p.KW) * p.PW + i % p.KW;
            p.tab_dy[i] = p.tab_dx[i] = 0;
        }
    } else {   // tap table in stride-parity plane order
        int n = 0;
        p.tab_new = 0;
        DAT_ENFORCE(ctx, p.KH * p.KW <= 32, "conv3d: %dx%d spatial kernel exceeds the 32-entry tap table", p.KH, p.KW);
        for (int py = 0; py < p.sh && py < p.KH; ++py)
            for (int px = 0; px < p.sw && px < p.KW; ++px) {
                bool first = true;
                for (int kh = py; kh < p.KH; kh += p.sh)
                    for (int kw = px; kw < p.KW; kw += p.sw) {
                        p.tab_tap[n] = kh * p.KW + kw;
                        p.tab_rowoff[n] = (kh / p.sh) * p.PW + (kw / p.sw);
                        p.tab_dy[n] = (short)py;
                        p.tab_dx[n] = (short)px;
                        if (first) p.tab_new |= 1u << n;
                        first = false;
                        ++n;
                    }
            }
        p.tab_n = n;
    }
    p.pw_magic = p.PW == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)p.PW - 1) / (unsigned)p.PW);
    p.n_cchunks = p.x3 ? p.Cin / Mma<DT>::CK / 2 * 3 : p.Cin / Mma<DT>::CK;   // bf16x3: [hi | hi | lo] against [W_hi | W_lo | W_hi]
    p.ablate = ctx->dbg_ablate;
    p.nblk_n = p.Cout_pad / BN;
    long long nblocks = (long long)p.frames * p.tiles_h * p.tiles_w * p.nblk_n;
    // split-K (chosen by plan_conv): fp32 partial sums in the ctx workspace, finished by splitk_finish_kernel
    {
        const int npatch_min = (p.KT > 1 ? p.KT - 1 : 1) * p.n_cchunks;
        int ks = ksplit < 1 ? 1 : (ksplit > npatch_min ? npatch_min : ksplit);
        p.ksplit = ks;
        p.part = nullptr;
        if (ks > 1) {
            const size_t bytes = (size_t)ks * p.frames * p.Ho * p.Wo * p.Cout * sizeof(float);
            int rc = dat_ensure_ws(ctx, bytes);
            if (rc != DAT_OK) return rc;
            p.part = (float*)ctx->ws;
        }
        nblocks *= ks;
    }
    DAT_ENFORCE(ctx, nblocks > 0 && nblocks < (1ll << 31), "conv3d: grid of %lld blocks unsupported", nblocks);
    DAT_ENFORCE(ctx, (long long)p.Ho * p.Wo * std::max(p.out_cs, p.Cout) * 4 < (1ll << 31),
                "conv3d: one output frame of %dx%dx%d exceeds the 2-GB range of the epilogue's 32-bit offsets", p.Ho, p.Wo, p.out_cs);
    p.nblocks = (unsigned)nblocks;
    p.ntiles = (unsigned)(nblocks / ((long long)p.ksplit * p.nblk_n));
    // Which blocks are neighbours in an XCD's queue (DAT_CONV_ORDER, default 0).  The ~64 resident blocks of an XCD read the weight slices
    // of every (cout block, split) they cover and the patches of their tiles through one 4-MB L2.  Order 1 (tile fastest) keeps ONE weight
    // slice resident and re-reads the input once per (cout block, split).  Measured: fabric reads -41 % over
    // the forward (res5's 512 -> 512 x 27 taps 2.09 -> 0.46 GB per launch, P2 7.8 -> 4.6 GB) and every such layer 2-4 % SLOWER -- the
    // sibling cout block no longer finds the patch in L2, and patch misses are waited for while weight misses are not.  Kept as a switch.
    p.order = ctx->dbg_order > 0;
    DAT_ENFORCE(ctx, p.tab_n % TPS == 0 && (TPS == 1 || p.tab_new == 1u), "conv3d: %d taps per step need one stride plane of a multiple of %d taps", TPS, TPS);
    size_t lds = (WD ? 0 : (size_t)2 * TPS * BN * ROWB) + (((size_t)p.PH * p.PW * PPITCH + 1023) & ~(size_t)1023);   // whole 1-KiB DMA pieces
    if (lds < 4 * 32 * (64 * 4 + 16)) lds = 4 * 32 * (64 * 4 + 16);                                   // epilogue staging slices
    lds += ctx->dbg_lds_pad;   // DEBUG: DAT_CONV_LDS_PAD=<bytes> lowers occupancy (blocks per CU) for experiments
    DAT_ENFORCE(ctx, lds <= 160 * 1024, "conv3d: LDS patch of %zu bytes exceeds 160 KiB (tile %dx%d, stride %dx%d)", lds,
                th, tw, p.sh, p.sw);
    auto kern = conv3d_igemm_kernel<DT, BN, BP, WAVES_N, TPS, WD, NTAP, ODT>;
    if (NTAP > 0) {   // what the unrolled variant assumes (the dispatcher only picks it for these shapes)
        DAT_ENFORCE(ctx, p.tab_n == (NTAP == 10 ? 9 : NTAP) && p.tab_new == 1u &&
                             (((size_t)p.PH * p.PW * 8 + 63) >> 6) <= (size_t)4 * patch_pieces_per_wave(NTAP, BP) && (NTAP == 10 || (size_t)p.PH * p.PW * PPITCH < 65536),
                    "conv3d: unrolled-tap variant on an unsupported shape (%d taps, patch %dx%d)", p.tab_n, p.PH, p.PW);
        for (int i = 0; i < p.tab_n; ++i) DAT_ENFORCE(ctx, p.tab_tap[i] == i, "conv3d: unrolled-tap variant needs taps in natural order");
    }
    {
        const int rc = dat_ensure_lds(ctx, (const void*)kern, 160 * 1024);
        if (rc != DAT_OK) return rc;
    }
#ifdef DAT_CONV_TRACE
    static unsigned long long* dbg = nullptr;
    if (!dbg) hipMalloc(&dbg, 128);
    hipMemsetAsync(dbg, 0, 128, st);
    p.dbg = dbg;
#endif
    hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(NTHREADS), lds, st, p);
#ifdef DAT_CONV_TRACE
    {
        unsigned long long h[16];
        hipStreamSynchronize(st);
        hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost);
        if (h[5]) fprintf(stderr, "TRACE BN=%d BP=%d blocks=%u steps/blk=%.1f | per step: reload %.0f wait %.0f barrier %.0f issue %.0f mma %.0f | shader clock %.0f MHz | per block: total %.0f epilogue %.0f cycles\n",
                          BN, BP, p.nblocks, (double)h[5] / p.nblocks, (double)h[0] / h[5], (double)h[1] / h[5], (double)h[2] / h[5],
                          (double)h[3] / h[5], (double)h[4] / h[5], 100.0 * (double)h[6] / (double)h[7], (double)h[8] / p.nblocks, (double)h[9] / p.nblocks);
    }
#endif
    if (p.ksplit > 1) {
        const size_t npos = (size_t)p.frames * p.Ho * p.Wo;
        const size_t tot = npos * (p.Cout >> 2);
        const int blocks = (int)std::min<size_t>((tot + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(splitk_finish_kernel<ODT>, dim3(blocks), dim3(256), 0, st, (const float*)p.part, p.ksplit, npos, p.Cout,
                           p.out_cs, p.scale, p.bias, p.res, p.res_mode, p.frames, p.Ho, p.Wo, p.relu, p.y, ODT == DAT_F32 && DT == DAT_BF16 ? p.y_split : nullptr, p.res2);
    }
    DAT_CHECK_LAUNCH(ctx, "conv3d_igemm");
    return DAT_OK;
}

}  // namespace

extern "C" {

int dat_conv3d_out_shape(const dat_conv_desc* d, int* Ho, int* Wo) {
    if (!d || d->stride_h < 1 || d->stride_w < 1) return DAT_ERR_ARG;
    *Ho = (d->H + 2 * d->pad_h - d->KH) / d->stride_h + 1;
    *Wo = (d->W + 2 * d->pad_w - d->KW) / d->stride_w + 1;
    return DAT_OK;
}

size_t dat_conv3d_packed_weight_bytes(const dat_conv_desc* d) {
    // (bf16x3: three bf16 chunks [W_hi | W_lo | W_hi] per logical 64-channel chunk)
    if (d->dtype == DAT_BF16X3) return (size_t)d->KT * d->KH * d->KW * cout_pad_of(d) * d->Cin * 3 * 2;
    return (size_t)d->KT * d->KH * d->KW * cout_pad_of(d) * d->Cin * dat_esize(d->dtype);
}

double dat_conv3d_flops(const dat_conv_desc* d, int Cin_real, int Cout_real) {
    int Ho, Wo;
    dat_conv3d_out_shape(d, &Ho, &Wo);
    const double oframes = d->out_tn > 0 && d->T > 0 ? (double)(d->frames / d->T) * d->out_tn : (double)d->frames;
    return 2.0 * Cout_real * Cin_real * d->KT * d->KH * d->KW * oframes * Ho * Wo;
}

static int conv3d_fwd_impl(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x, const void* w_packed,
                           const float* scale, const float* bias, const void* residual, void* y, void* y_split, const void* addend = nullptr);

int dat_conv3d_fwd(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x, const void* w_packed,
                   const float* scale, const float* bias, const void* residual, void* y) {
    return conv3d_fwd_impl(ctx, s, d, x, w_packed, scale, bias, residual, y, nullptr);
}

int dat_conv3d_fwd_x3(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x_split, const void* w_packed,
                      const float* scale, const float* bias, const void* residual, void* y, void* y_split) {
    DAT_ENFORCE(ctx, d && d->dtype == DAT_BF16X3, "conv3d_fwd_x3: the descriptor's dtype must be DAT_BF16X3");
    DAT_ENFORCE(ctx, !y_split || (d->out_cstride % 64 == 0 && d->Cout == d->out_cstride),
                "conv3d_fwd_x3: a split output needs Cout == out_cstride, a multiple of 64 (got %d / %d)", d->Cout, d->out_cstride);
    return conv3d_fwd_impl(ctx, s, d, x_split, w_packed, scale, bias, residual, y, y_split);
}

int dat_conv3d_fwd_sum_mask(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x, const void* w_packed,
                            const float* scale, const float* bias, const void* addend, const void* mask, void* y) {
    DAT_ENFORCE(ctx, d && d->res_mode == 4, "conv3d_fwd_sum_mask: the descriptor's res_mode must be 4");
    DAT_ENFORCE(ctx, addend && mask, "conv3d_fwd_sum_mask: null addend / mask");
    return conv3d_fwd_impl(ctx, s, d, x, w_packed, scale, bias, mask, y, nullptr, addend);
}

static int conv3d_fwd_impl(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const void* x, const void* w_packed,
                           const float* scale, const float* bias, const void* residual, void* y, void* y_split, const void* addend) {
    DAT_ENFORCE(ctx, d && x && w_packed && y, "conv3d_fwd: null argument");
    DAT_ENFORCE(ctx, d->dtype == DAT_F32 || d->dtype == DAT_BF16 || d->dtype == DAT_BF16X3, "conv3d_fwd: bad dtype %d", d->dtype);
    const bool x3 = d->dtype == DAT_BF16X3;
    DAT_ENFORCE(ctx, !x3 || DAT_H16_FORMAT == 0, "conv3d_fwd: DAT_BF16X3 needs the bf16 build of the library (this one holds IEEE half in its 16-bit tensors)");
    DAT_ENFORCE(ctx, !x3 || weights_direct(ctx, d), "conv3d_fwd: the bf16x3 mode runs on the weights-direct kernel variants (DAT_CONV_WD)");
    DAT_ENFORCE(ctx, d->Cin % 64 == 0, "conv3d_fwd: Cin (channel stride) %d must be a multiple of 64", d->Cin);
    DAT_ENFORCE(ctx, d->Cout % 4 == 0 && d->out_cstride % 4 == 0 && d->out_cstride >= d->Cout,
                "conv3d_fwd: Cout %d / out_cstride %d must be multiples of 4", d->Cout, d->out_cstride);
    DAT_ENFORCE(ctx, d->frames % d->T == 0, "conv3d_fwd: frames %d not a multiple of T %d", d->frames, d->T);
    DAT_ENFORCE(ctx, d->res_mode >= 0 && d->res_mode <= 4, "conv3d_fwd: res_mode %d", d->res_mode);
    DAT_ENFORCE(ctx, d->res_mode != 4 || (!x3 && !y_split && addend), "conv3d_fwd: res_mode 4 (sum + mask) goes through dat_conv3d_fwd_sum_mask, bf16 / fp32 only");
    DAT_ENFORCE(ctx, d->res_mode == 0 || residual, "conv3d_fwd: res_mode %d needs a residual pointer", d->res_mode);
    // full-length outputs need "same" temporal padding; an explicit output-frame window may use any pad_t (taps that
    // fall outside [0, T) read zeros) -- e.g. KT == T, pad_t == 0, window {0}: a 1x1 conv over time-moved-to-channels
    DAT_ENFORCE(ctx, d->pad_t * 2 + 1 == d->KT || d->out_tn > 0,
                "conv3d_fwd: temporal pad %d must be (KT-1)/2 for KT %d unless out_t0/out_tn select the output frames",
                d->pad_t, d->KT);
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const char*)x; p.w = (const char*)w_packed; p.scale = scale; p.bias = bias;
    p.res = (const char*)residual; p.res2 = (const char*)addend; p.y = (char*)y; p.y_split = (char*)y_split;
    p.zeros = (const char*)ctx->zeros;
    p.clk = ctx->prof_enabled ? (unsigned long long*)((char*)ctx->zeros + 256) : nullptr;
    p.in_lo = d->in_tn > 0 ? d->in_t0 : 0;
    p.in_hi = d->in_tn > 0 ? d->in_t0 + d->in_tn : d->T;
    DAT_ENFORCE(ctx, p.in_lo >= 0 && p.in_hi <= d->T, "conv3d_fwd: non-zero input frames [%d, %d) outside T %d", p.in_lo, p.in_hi, d->T);
    p.ot0 = d->out_tn > 0 ? d->out_t0 : 0;
    p.otn = d->out_tn > 0 ? d->out_tn : d->T;
    DAT_ENFORCE(ctx, p.ot0 >= 0 && p.ot0 + p.otn <= d->T, "conv3d_fwd: output frames [%d, %d) outside T %d", p.ot0,
                p.ot0 + p.otn, d->T);
    p.frames = d->frames / d->T * p.otn; p.T = d->T; p.H = d->H; p.W = d->W; p.Cin = x3 ? 2 * d->Cin : d->Cin;   // (x3: pixel pitch of the hi / lo split tensor)
    p.x3 = x3;
    dat_conv3d_out_shape(d, &p.Ho, &p.Wo);
    DAT_ENFORCE(ctx, p.Ho > 0 && p.Wo > 0, "conv3d_fwd: empty output %dx%d", p.Ho, p.Wo);
    DAT_ENFORCE(ctx, d->res_mode != 2 || (p.Ho % 2 == 0 && p.Wo % 2 == 0), "conv3d_fwd: res_mode 2 needs even output dims");
    p.Cout = d->Cout; p.out_cs = d->out_cstride; p.Cout_pad = cout_pad_of(d);
    p.KT = d->KT; p.KH = d->KH; p.KW = d->KW; p.sh = d->stride_h; p.sw = d->stride_w;
    p.pt = d->pad_t; p.ph = d->pad_h; p.pw = d->pad_w; p.relu = d->relu; p.res_mode = d->res_mode;
    hipStream_t st = (hipStream_t)s;

    const bool small_n = d->Cout <= 64;
    // ---- plan: positions per block (128 | 256) and split-K factor from a small makespan model ----------------
    // A block runs steps = ceil(npatch/ks) * taps tap-steps; 2 blocks share a CU (512 slots).  Measured on MI355X:
    // a BP=128 step costs ~1.35 us, a BP=256 step ~2.15 us (2x the work).  The grid runs in "rounds" of 512 blocks;
    // with few rounds the last partial round costs a full one.  Split-K adds an fp32 partial round trip.
    const int force_bp = ctx->force_bp, force_ks = ctx->force_ks;
    const int ck = d->dtype == DAT_F32 ? 32 : 64;
    const int ncc = x3 ? d->Cin / 64 * 3 : d->Cin / ck;
    const int npatch = d->KT * ncc, npatch_min = (d->KT > 1 ? d->KT - 1 : 1) * ncc;
    const int ntaps = d->KH * d->KW;
    const long long nbn = p.Cout_pad / (small_n ? 64 : 128);
    int bp = 128, ksplit = 1;
    {
        double best = 1e30;
        const bool deep_1x1 = (ntaps == 1 && ncc >= 16);
        for (int cand_bp = 128; cand_bp <= 256; cand_bp += 128) {
            if (cand_bp == 256 && (small_n || ntaps == 1)) continue;   // the 64-channel and 1x1 variants do not profit
            // strided 3x3: the dense-patch variant exists for 128-position tiles only and beats the 256-position table-driven loop
            // (res4_0_branch2a 0.128 -> 0.081 ms, res3_0_branch2a 0.086 -> 0.081, cold caches)
            if (cand_bp == 256 && !force_bp && ctx->dbg_ntap && !(ctx->dbg_ntap & 4) && d->KH == 3 && d->KW == 3 && d->stride_h == 2 && d->stride_w == 2) continue;
            if (force_bp && cand_bp != force_bp && !(force_bp == 256 && (small_n || ntaps == 1))) continue;
            const int lg = cand_bp == 256 ? 8 : 7;
            const TileChoice tc = choose_tile(p.Ho, p.Wo, lg, p.sh, p.sw, p.KH, p.KW, ctx->dbg_tw_log2);
            long long tiles = cdiv_ll(p.Ho, 1ll << tc.th_log2) * cdiv_ll(p.Wo, 1ll << tc.tw_log2) * p.frames;
            if (weights_direct(ctx, d)) {   // (the dispatcher's rule below: linear tiles when they save >= 10 %)
                const long long tl = linear_tiles(ctx, d, p, cand_bp);
                if (tl > 0 && tl * 10 <= tiles * 9) tiles = tl;
            }
            const int ks_max = (ntaps > 1 || deep_1x1) ? (deep_1x1 ? 8 : 4) : 1;
            for (int ks = 1; ks <= ks_max && ks <= npatch_min; ++ks) {
                if (force_ks && ks != std::min(force_ks, npatch_min)) continue;
                const double nblk = (double)tiles * nbn * ks;
                const double slots = 512.0;
                double rounds = nblk / slots;
                if (rounds <= 3.0) rounds = ceil(rounds); else rounds += 0.5;
                const double steps = ceil((double)npatch / ks) * ntaps;
                // (stride-2 3x3 convs run the table-driven loop with four stride-parity patches per chunk: the 256-position tile pays
                //  more per step there -- tools/tune_plan.py, round 2: res4_0_branch2a 0.114 ms at 256 positions vs 0.089 at 128)
                const bool strided_taps = ntaps > 1 && (d->stride_h > 1 || d->stride_w > 1);
                const double step_us = cand_bp == 256 ? (strided_taps ? 2.8 : 2.15) : 1.35;
                const double fixed_us = 2.0 + 0.6 * ceil((double)npatch / ks);      // epilogue + exposed patch loads
                double t = rounds * (steps * step_us + fixed_us);
                if (ks > 1) t += 4.0 + 2.0 * ks * (double)p.frames * p.Ho * p.Wo * d->Cout * 4.0 / 3.0e6;  // us @3 TB/s
                if (t < best) { best = t; bp = cand_bp; ksplit = ks; }
            }
        }
    }
    const bool big = bp == 256;
    int tag = (small_n ? 64 : 128) * 10000 + bp * 10 + d->dtype;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // (no event pairs inside a stream capture: an event recorded there becomes a graph node, and timing it later fails with
    //  hipErrorInvalidHandle -- a sticky error the host framework then reports at an unrelated call)
    hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
    if (ctx->prof_enabled && ctx->prof_n < ctx->prof_cap && hipStreamIsCapturing(st, &cap_st) == hipSuccess &&
        cap_st == hipStreamCaptureStatusNone) {
        e0 = ctx->prof_ev[2 * ctx->prof_n];
        e1 = ctx->prof_ev[2 * ctx->prof_n + 1];
        hipEventRecord(e0, st);
    }
    int rc;
    const int tps3 = ctx->dbg_tps3;
    // thin layers (<= 64 output channels, dense 3x3 spatial taps): 3 taps per step
    const bool thin3 = tps3 && small_n && !big && d->stride_h == 1 && d->stride_w == 1 && d->KH == 3 && d->KW == 3 && !weights_direct(ctx, d);
    // big-tile kernel: one block per CU at a time, so the grid has to fill the chip several times and evenly (FPN P2: 2016 blocks on
    // 256 CUs = 7.9 rounds; P3's 528 blocks would run 3 rounds for 2.06 of work: 0.45 ms against 0.34 ms with the generic kernel)
    long long bt_blocks = 0;
    if (bt_eligible(ctx, d)) bt_tile_twl(p, &bt_blocks);
    const long long ncu = ctx_num_cu(ctx), bt_rounds = cdiv_ll(bt_blocks, ncu);
    const bool bt_fits = ctx->dbg_bt >= 2 ? bt_blocks * 2 >= 3 * ncu : (bt_blocks * 100 >= (long long)ctx->dbg_bt_min * ncu && bt_blocks * 100 >= bt_rounds * ncu * 95);   // (3.9: conv_rpn_fpn2 at 4 clips is 1 008 blocks on 256 CUs)
    const bool mask_mode = d->res_mode >= 3;     // the masking combines (3: mask, 4: in-place sum + mask) live in the generic kernel and the lw / ks 1x1 kernels
    if (bt_fits && ksplit == 1 && !force_bp && !force_ks && !mask_mode) {
        tag = 256 * 10000 + 2560 + d->dtype;
        rc = launch_bt(ctx, st, p);
    } else if (pw256_eligible(ctx, d) && !force_bp && !force_ks && !mask_mode) {
        tag = 256 * 10000 + 320 + d->dtype;     // (256 channels x 32 positions per wave: the weights-stationary 1x1 kernel)
        rc = launch_pw256(ctx, st, p);
    } else if (pwks_eligible(ctx, d) && !force_bp && !force_ks && !x3) {       // (its epilogue knows the masking combine too)
        tag = 256 * 10000 + 340 + d->dtype;     // (the K-streaming 1x1 kernel: 256 positions x 256 channels per block)
        rc = launch_pwks(ctx, st, p);
    } else if (pwlw_eligible(ctx, d) && !force_bp && !force_ks && d->res_mode != 3) {   // (knows the sum + mask combine; plain mask layers -- K >= 512 data
                                                                                        //  gradients of `branch2c` on small maps -- measured faster on the generic kernel)
        tag = 256 * 10000 + 330 + d->dtype;     // (the weights-in-LDS 1x1 kernel: 32 positions per wave tile)
        rc = launch_pwlw(ctx, st, p, d);
    } else if (ws64_eligible(ctx, d) && !force_bp && !force_ks && !mask_mode) {
        tag = 64 * 10000 + 9990 + d->dtype;    // ("999 positions": the persistent weights-stationary kernel)
        rc = launch_ws64(ctx, st, p);
    } else if (thin3) {
        tag += 3;   // (dtype digit + 3: the 3-taps-per-step variant)
        rc = d->dtype == DAT_BF16 ? launch_conv<DAT_BF16, 64, 128, 1, 3>(ctx, st, p, 7, ksplit) : launch_conv<DAT_F32, 64, 128, 1, 3>(ctx, st, p, 7, ksplit);
    } else if (weights_direct(ctx, d)) {   // weights straight into the MFMA registers (all 128-channel tiles; 64-channel ones at DAT_CONV_WD=2)
        // unrolled-tap variants: dense 3x3 (stride 1) and 1x1 (any stride: one tap, one plane); the patch must fit the per-wave
        // piece registers (tile shapes with very long rows fall back to the table-driven loop).  1x1 layers with a large output
        // grid are bandwidth-bound and do better with the leaner table-driven loop (one block more per CU), measured.
        const TileChoice tc = choose_tile(p.Ho, p.Wo, big ? 8 : 7, p.sh, p.sw, p.KH, p.KW, ctx->dbg_tw_log2);
        const long long prow = ((1ll << tc.th_log2) + (p.KH - 1) / p.sh) * ((1ll << tc.tw_log2) + (p.KW - 1) / p.sw);
        const bool fits = ((prow * 8 + 63) >> 6) <= 4 * patch_pieces_per_wave(9, big ? 256 : 128);
        const bool pw_small = (long long)p.frames * p.Ho * p.Wo <= 65536;
        // strided 3x3: ONE dense patch of (2*TH+1) x (2*TW+1) cells serves all 9 taps (the table-driven loop stages four stride-parity
        // patches per channel chunk and synchronises around each); 128-position tiles only (the patch is 70 KiB: two blocks per CU)
        const long long drow = (((1ll << tc.th_log2) - 1) * p.sh + p.KH) * (((1ll << tc.tw_log2) - 1) * p.sw + p.KW);
        const bool dense2 = ctx->dbg_ntap && !big && d->KH == 3 && d->KW == 3 && d->stride_h == 2 && d->stride_w == 2 && !(ctx->dbg_ntap & 4) &&
                            ((drow * 8 + 63) >> 6) <= 4 * 19;
        const int ntapv = dense2 ? 10 : !ctx->dbg_ntap || !fits ? 0 : (d->KH == 3 && d->KW == 3 && d->stride_h == 1 && d->stride_w == 1) ? 9 :
                          (d->KH == 1 && d->KW == 1 && (pw_small || (ctx->dbg_ntap & 2))) ? 1 : 0;
        // small maps (RoI heads): linear position tiling when it saves >= 10 % of the tiles (see launch_conv)
        bool lin = false, lin320 = false;
        const long long tlin = ntapv == 9 ? linear_tiles(ctx, d, p, big ? 256 : 128) : 0;
        if (tlin > 0) {
            const long long total = (long long)p.frames * p.Ho * p.Wo;
            const long long t2d = (long long)p.frames * cdiv_ll(p.Ho, 1ll << tc.th_log2) * cdiv_ll(p.Wo, 1ll << tc.tw_log2);
            lin = tlin * 10 <= t2d * 9;
            // One block per CU runs a tap step in ~1.1 us, two co-resident ones in ~2.15 us each, so a grid just above the CU count
            // costs a whole second block lifetime (keypoint head, 100 x 14 x 14 maps: 83 maps = 256 blocks 0.081 ms, 84 maps = 260
            // blocks 0.119 ms).  320-position tiles bring such a grid back under one block per CU.
            const long long ncu320 = ctx_num_cu(ctx), blk256 = tlin * nbn * ksplit, blk320 = cdiv_ll(total, 320) * nbn * ksplit;
            // (opt-in, DAT_CONV_LINEAR=5: one clip at a time +2.9 % (170.6 -> 175.5 clips/s), but with four clips in flight -2 % (221.8 -> 217.5):
            //  a 293-register block per CU on 248 CUs leaves no room for the other clips' kernels to run beside it)
            lin320 = lin && d->KT == 1 && big && !small_n && (ctx->dbg_linear & 4) && blk256 > ncu320 && blk320 <= ncu320 &&
                     ((((320 + 2 * (p.Wo + 1) + 1) * 8 + 63) >> 6) <= 4 * patch_pieces_per_wave(9, 320));
        }
#define DAT_WD_LAUNCH(DT_, BN_, WN_, ODT_) (ntapv == 10 ? launch_conv<DT_, BN_, 128, WN_, 1, 1, 10, ODT_>(ctx, st, p, 7, ksplit) : ntapv == 9 ? (big ? launch_conv<DT_, BN_, 256, WN_, 1, 1, 9, ODT_>(ctx, st, p, 8, ksplit, lin) : launch_conv<DT_, BN_, 128, WN_, 1, 1, 9, ODT_>(ctx, st, p, 7, ksplit, lin)) \
                            : ntapv == 1 ? (big ? launch_conv<DT_, BN_, 256, WN_, 1, 1, 1, ODT_>(ctx, st, p, 8, ksplit) : launch_conv<DT_, BN_, 128, WN_, 1, 1, 1, ODT_>(ctx, st, p, 7, ksplit)) \
                            : (big ? launch_conv<DT_, BN_, 256, WN_, 1, 1, 0, ODT_>(ctx, st, p, 8, ksplit) : launch_conv<DT_, BN_, 128, WN_, 1, 1, 0, ODT_>(ctx, st, p, 7, ksplit)))
        if (x3) rc = small_n ? DAT_WD_LAUNCH(DAT_BF16, 64, 1, DAT_F32) : DAT_WD_LAUNCH(DAT_BF16, 128, 2, DAT_F32);
        else if (lin320) rc = d->dtype == DAT_BF16 ? launch_conv<DAT_BF16, 128, 320, 2, 1, 1, 9>(ctx, st, p, 9, ksplit, true)
                                              : launch_conv<DAT_F32, 128, 320, 2, 1, 1, 9>(ctx, st, p, 9, ksplit, true);
        else if (small_n) rc = d->dtype == DAT_BF16 ? DAT_WD_LAUNCH(DAT_BF16, 64, 1, DAT_BF16) : DAT_WD_LAUNCH(DAT_F32, 64, 1, DAT_F32);
        else rc = d->dtype == DAT_BF16 ? DAT_WD_LAUNCH(DAT_BF16, 128, 2, DAT_BF16) : DAT_WD_LAUNCH(DAT_F32, 128, 2, DAT_F32);
#undef DAT_WD_LAUNCH
    } else if (d->dtype == DAT_BF16) {
        if (big)
            rc = small_n ? launch_conv<DAT_BF16, 64, 256, 1>(ctx, st, p, 8, ksplit) : launch_conv<DAT_BF16, 128, 256, 2>(ctx, st, p, 8, ksplit);
        else
            rc = small_n ? launch_conv<DAT_BF16, 64, 128, 1>(ctx, st, p, 7, ksplit) : launch_conv<DAT_BF16, 128, 128, 2>(ctx, st, p, 7, ksplit);
    } else {
        if (big)
            rc = small_n ? launch_conv<DAT_F32, 64, 256, 1>(ctx, st, p, 8, ksplit) : launch_conv<DAT_F32, 128, 256, 2>(ctx, st, p, 8, ksplit);
        else
            rc = small_n ? launch_conv<DAT_F32, 64, 128, 1>(ctx, st, p, 7, ksplit) : launch_conv<DAT_F32, 128, 128, 2>(ctx, st, p, 7, ksplit);
    }
    if (e1) {
        hipEventRecord(e1, st);
        ctx->prof_flops[ctx->prof_n] = 2.0 * d->Cout * d->Cin * d->KT * d->KH * d->KW * (double)p.frames * p.Ho * p.Wo;
        ctx->prof_tag[ctx->prof_n] = tag;
        ctx->prof_n++;
    }
    return rc;
}

int dat_conv3d_persistent_share(dat_ctx* ctx, int percent) {
    if (!ctx) return DAT_ERR_ARG;
    DAT_ENFORCE(ctx, percent >= 1 && percent <= 100, "conv3d_persistent_share: %d percent out of range", percent);
    ctx->dbg_persist_pct = percent;
    return DAT_OK;
}

int dat_conv3d_tune_plan(dat_ctx* ctx, int positions_per_block, int ksplit) {
    if (!ctx) return DAT_ERR_ARG;
    DAT_ENFORCE(ctx, (positions_per_block == 0 || positions_per_block == 128 || positions_per_block == 256) && ksplit >= 0 && ksplit <= 8,
                "conv3d_tune_plan: positions per block %d / split-K %d out of range", positions_per_block, ksplit);
    ctx->force_bp = positions_per_block;
    ctx->force_ks = ksplit;
    return DAT_OK;
}

}  // extern "C"
