// Internal header of the conv translation units (conv3d_igemm.hip: the generic implicit-GEMM kernel, its plan and the dispatcher;
// conv_special.hip: the weights-stationary and big-tile kernels; conv_pack.hip: weight packing).  Not part of the C ABI.
#ifndef DAT_CONV_INTERNAL_H
#define DAT_CONV_INTERNAL_H

#include <math.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "dat_common.h"

namespace dat_conv __attribute__((visibility("hidden"))) {

constexpr int ROWB = 128;   // bytes per weight LDS row (one 128-B line of channels), XOR-swizzled
constexpr int PPITCH = 128; // patch row pitch: one 128-B line per pixel, lane-linear LDS-DMA image, XOR-swizzled like the weights
constexpr int NTHREADS = 256;
#ifndef DAT_KT_ROTATE
#define DAT_KT_ROTATE 1
#endif

// 1-KiB pieces of the patch one wave of the unrolled-tap kernels copies (a register of source offsets each): 2-D tiles need
// (16+2)x(16+2) rows = 41 pieces at 256 positions; the linear strips of wide maps (res4: 256 + 2 * 85 + 1 rows) up to 54
constexpr int patch_pieces_per_wave(int ntap, int bp) { return ntap == 10 ? 19 : bp >= 256 ? 14 : 10; }

struct ConvParams {
    const char* x;
    const char* w;
    const float* scale;
    const float* bias;
    const char* res;
    const char* res2;          // res_mode 4: the addend (the other gradient contribution; may be y itself: every element is read and written by one thread)
    char* y;
    char* y_split;              // bf16x3 mode only (ODT fp32): also write the hi / lo bf16 split of y (pixel pitch 2 * out_cs bf16, dat_split_bf16x2's layout) -- the
                                // next conv then needs no split pre-pass; NULL = off
    unsigned long long* dbg;    // DAT_CONV_TRACE builds only: per-phase cycle sums
    unsigned long long* clk;    // profiling only (dat_prof_enable): [0] += shader cycles, [1] += 100-MHz ticks per block
    const char* zeros;          // >= 16 zero bytes (what a halo lane of the patch LDS-DMA fetches)
    int frames, T, H, W, Cin;   // frames = OUTPUT frames (clips * otn)
    int ot0, otn;               // output frames per clip: t in [ot0, ot0 + otn)
    int in_lo, in_hi;           // input frames outside [in_lo, in_hi) of the clip are known to be zero: their temporal taps are skipped
    int Ho, Wo, Cout, out_cs, Cout_pad;
    int KT, KH, KW, sh, sw, pt, ph, pw;
    int relu, res_mode;
    int th_log2, tw_log2;     // output tile = 2^th x 2^tw positions
    int tiles_h, tiles_w;
    int PH, PW;               // patch rows/cols (LDS rows = PH*PW)
    int psh, psw;             // patch sampling step in the input (= conv stride; 1 for the dense stride-2 patch of the NTAP = 10 variant)
    int rsh, rsw;             // patch rows / columns between neighbouring output positions (1; 2 for the dense stride-2 patch)
    int lin_h, lin_w;         // > 0: LINEAR position tiling of small maps (see launch_conv): the real map size; H / W / Ho / Wo then describe a 1 x N strip
    int tile_w;               // output columns between neighbouring tiles
    int lin_zero_row;         // patch row that is all zeros (-1: none): the B fragments of taps that fall outside a map read it
    // spatial tap schedule of one (kt, channel chunk): taps grouped by stride-parity plane, so that every plane is a
    // dense (tile + halo/stride) patch whose rows are read consecutively (stride-2 convs: 4 small patches)
    int tab_n;
    int tab_tap[32];          // kh*KW + kw (weight tap index)
    int tab_rowoff[32];       // LDS row offset of this tap inside the plane patch
    short tab_dy[32], tab_dx[32];  // input offset of the plane's patch cell (0,0) relative to (ih0, iw0)
    unsigned tab_new;         // bit i: entry i starts a new plane (patch reload)
    unsigned pw_magic;        // ceil(2^32 / PW): row / PW == umulhi(row, pw_magic) for row, PW < 2^16; 0 when PW == 1 (2^32 does
                              // not fit: a 1-wide patch of a KW == 1 conv made every row decode to patch row 0)
    int n_cchunks;            // Cin / CK
    int ksplit;               // split-K over the (kt, channel-chunk) sequence; > 1 => fp32 partials to `part`
    float* part;              // [ksplit][frames*Ho*Wo][Cout] fp32 (split-K only)
    int ablate;               // DEBUG (DAT_CONV_ABLATE): 1 skip patch reloads, 2 skip weight streaming
    int nblk_n;               // Cout_pad / BN
    unsigned nblocks;
    unsigned ntiles;          // position tiles (frames x tiles_h x tiles_w)
    int x3;                   // bf16x3 mode: x is the hi / lo split tensor (pixel pitch Cin = 2 x logical channels), channel chunk cc of the K loop reads
                              // source line (cc / 3) * 2 + (cc % 3 == 2); output / residual are fp32 (template parameter ODT)
    int order;                // block order inside an XCD's queue: 0 = (split, cout block) fastest -- the blocks of one tile share its input patch;
                              // 1 = tile fastest -- the resident blocks share ONE (cout block, split) weight slice (layers whose weights exceed the L2)
};

template <int DT> struct Mma;
template <> struct Mma<DAT_BF16> {
    static constexpr int CK = 64;  // channels per 128-B row
    __device__ static __forceinline__ void step(const uint4& a, const uint4& b, f32x16_t& c) {
        c = DAT_MFMA16(a, b, c);
    }
};
template <> struct Mma<DAT_F32> {
    static constexpr int CK = 32;
    __device__ static __forceinline__ void step(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// residual combine of the fused epilogue: modes 1 / 2 add (Sum shortcut, FPN top-down map); mode 3 MASKS -- out = res > 0 ? v : 0, the
// ReLU backward of the conv's INPUT blob fused into the data-gradient conv (round 3: `res` is the forward input x = relu(...) of the
// conv whose data gradient the launch computes; training.py bwd_Conv)
__device__ __forceinline__ float res_combine(float v, float r, int mode) { return mode == 3 ? (r > 0.f ? v : 0.f) : v + r; }
// mode 4 (round 6): SUM + MASK -- out = m > 0 ? v + old : 0, where `old` is the addend operand (the other contribution to the gradient of a
// residual block's output; it may be the output tensor itself) and `m` the residual operand (that block output y = relu(...)): the data-gradient
// conv of the LAST reader of y finishes the sum and applies y's ReLU backward in one epilogue (dat_conv3d_fwd_sum_mask; training.py bwd_Conv)
__device__ __forceinline__ float res_combine4(float v, float old, float m) { return m > 0.f ? v + old : 0.f; }

__device__ __forceinline__ int swz(int row, int slot) { return (row * ROWB) + (((slot ^ (row >> 1)) & 7) << 4); }


// compile-time index sequence for the hand-scheduled loops (`#pragma unroll` is refused for bodies of this size, and immediates /
// register-ring slots need constant indices)
template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}


inline int cout_pad_of(const dat_conv_desc* d) {
    const int bn = d->Cout <= 64 ? 64 : 128;
    return (d->Cout + bn - 1) / bn * bn;
}

// packed-weight layout of a layer: the 128-channel tile variants read A fragments straight from global memory (WD)
inline bool weights_direct(const dat_ctx* ctx, const dat_conv_desc* d) { return ctx->dbg_wd && (d->Cout > 64 || ctx->dbg_wd >= 2); }

// conv_special.hip
int ctx_num_cu(dat_ctx* ctx);
bool ws64_eligible(const dat_ctx* ctx, const dat_conv_desc* d);
int launch_ws64(dat_ctx* ctx, hipStream_t st, const ConvParams& cp);
bool pw256_eligible(const dat_ctx* ctx, const dat_conv_desc* d);
int launch_pw256(dat_ctx* ctx, hipStream_t st, const ConvParams& cp);
bool pwlw_eligible(const dat_ctx* ctx, const dat_conv_desc* d);
int launch_pwlw(dat_ctx* ctx, hipStream_t st, const ConvParams& cp, const dat_conv_desc* d);
bool pwks_eligible(const dat_ctx* ctx, const dat_conv_desc* d);
int launch_pwks(dat_ctx* ctx, hipStream_t st, const ConvParams& cp);
bool bt_eligible(const dat_ctx* ctx, const dat_conv_desc* d);
int bt_tile_twl(const ConvParams& p, long long* nblocks);
int launch_bt(dat_ctx* ctx, hipStream_t st, ConvParams& p);

}  // namespace dat_conv

#endif
