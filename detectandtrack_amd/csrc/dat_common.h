// Shared device/host helpers for libdat_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/dat_hip.h"

struct dat_ctx {
    int device;
    std::string last_error;
    // optional per-launch timing of the conv kernel (bench.py roofline leg)
    int prof_enabled;
    hipEvent_t* prof_ev;   // pairs
    int prof_cap, prof_n;
    double* prof_flops;    // algorithmic flops per recorded launch
    int* prof_tag;         // kernel variant tag per recorded launch
    // scratch owned by the ctx, grown on demand (proposal path)
    void* ws;
    size_t ws_bytes;
    std::vector<void*> ws_retired;   // outgrown scratch buffers: captured hipGraphs may still replay launches that point at them
    int ws_generation;               // bumped at every growth (dat_ws_info)
    void* zeros;           // 512 B in HBM: [0,256) zeros (conv patch loader: source of out-of-frame halo lanes),
                           // [256,272) profiling clock counters of the conv kernel
    void* util_stream;     // private non-blocking hipStream_t for the context's own small transfers
    void* pinned;          // 256 B of pinned host memory (their host side)
    // Launch-plan state of THIS context (no process globals: N contexts / N devices per process are independent,
    // SURVEY.md 8b "thread-safe per dat_ctx").  The debug knobs are read from the environment once, at dat_ctx_create.
    int force_bp, force_ks;                 // dat_conv3d_tune_plan / DAT_CONV_BP, DAT_CONV_KSPLIT; 0 = makespan model
    int dbg_tw_log2, dbg_ablate, dbg_lds_pad, dbg_tps3;   // DAT_CONV_TW_LOG2 (-1 = off), DAT_CONV_ABLATE, DAT_CONV_LDS_PAD, DAT_CONV_TPS
    int dbg_pack_simple;                    // DAT_PACK_SIMPLE (default 0): element-wise weight packing instead of the LDS-tiled kernel
    int dbg_linear;                         // DAT_CONV_LINEAR (default 1): linear position tiling of small maps (RoI-head 3x3 convs on 14 x 14 maps); 5 = also the 320-position one-block-per-CU tiles
    int dbg_order;                          // DAT_CONV_ORDER (default 0): block order of the implicit-GEMM conv inside an XCD's queue, 0 patch-sharing, 1 weight-stationary (less fabric traffic, slower)
    int dbg_persist_pct;                    // DAT_PERSIST_PCT (default 100): percent of the CUs the persistent HBM-bound conv kernels take (experiment)
    int dbg_bt_min;                         // DAT_CONV_BT_MIN (default 390): the big-tile kernel takes grids of at least this many hundredths of a round of the CUs
    int dbg_bt;                             // DAT_CONV_BT (default 1 = on for evenly filling grids of >= 3.9 rounds; 0 = off): big-tile (256 x 256, one wave per SIMD) kernel for 3x3 layers with 256-channel-multiple outputs and an evenly filling grid; 2 = for every grid of >= 1.5 blocks per CU (tests)
    int dbg_ws64;                           // DAT_CONV_WS64 (default 1): weights-stationary persistent kernel for 3x3 64 -> 64 bf16 layers
    int dbg_pwks;                           // DAT_CONV_PWKS (default 8: K >= 512): K-streaming 1x1 kernel for layers with at least this many 64-channel chunks (0 = off)
    int dbg_pwlw;                           // DAT_CONV_PWLW (default 1): weights-in-LDS persistent kernel for HBM-bound 1x1 layers (K <= 512, weights of a cout part <= 128 KB)
    int dbg_wgrad_direct;                   // DAT_WGRAD_DIRECT (default 1): bf16 weight gradients straight from the NDHWC tensors (transposing LDS reads), no re-pack passes
    int dbg_ablate_wgrad;                   // DAT_WGRAD_ABLATE (default 0): DEBUG timing ablations of the nine-tap weight-gradient kernel (wrong results)
    int dbg_wgrad_sub;                      // DAT_WGRAD_SUB (default 2): 2 = eight-wave blocks of the nine-tap weight-gradient kernel (two K ranges per block, added through LDS)
    int dbg_wgrad_ilv;                      // DAT_WGRAD_ILV (default 1): 1 = LDS-DMA pieces of the next-but-one chunk issued between the MFMA groups of the current one
    int dbg_wgrad_xcd;                      // DAT_WGRAD_XCD (default 0): XCD-aware block order of wgrad_dma9_kernel (consecutive logical blocks on one XCD) -- measured neutral (round 6: R-50 iteration 20.22 vs 20.39 ms, R-18 14.60 vs 14.56)
    int dbg_pw_xcd;                         // DAT_CONV_PW_XCD (default 1): the cout parts / cout blocks of the 1x1 kernels (conv1x1_lw / conv1x1_ks) that read one input tile share an XCD
    int dbg_wgrad_dma;                      // DAT_WGRAD_DMA (default 1): nine-tap weight gradient with LDS-DMA operand staging (three stages) instead of register staging
    int dbg_wgrad_ks;                       // DAT_WGRAD_KS (default 0 = heuristic): forced K split of the nine-tap direct weight-gradient kernel
    int dbg_wgrad_pw;                       // DAT_WGRAD_PW (default 1): eight-wave 64 K-accumulator kernel for pointwise weight gradients (0: the 128 x 128 per-tap kernel; 10 / 20 / 40: forced tile shape)
    int dbg_kps_sep;                        // DAT_KPS_DECODE_SEP (default 1): separable heatmap decode (horizontal pass per 64-column strip in LDS); 0 = the per-pixel 4 x 4 kernel
    int dbg_ws_poison;                      // DAT_WS_POISON (default 0): a grown scratch buffer is filled with 0xFF (tests: no kernel may rely on fresh hipMalloc pages reading as zero)
    int dbg_roi_fold;                       // DAT_ROI_BWD_FOLD (default 1): RoIAlign backward folds a bin's samples into one weight per distinct pixel before the atomics
    int num_cu;                             // compute units of the device (persistent-kernel grids)
    int dbg_ntap;                           // DAT_CONV_NTAP (default 1): unrolled-tap variants of the WD kernels (3x3 stride 1, 1x1)
    int dbg_wd;                             // DAT_CONV_WD (default 2): tiles read their weights straight from global memory (1: only the 128-channel ones)
    // kernels whose dynamic-LDS limit was already raised on this context's device (the attribute is per device)
    std::unordered_set<const void*> lds_attr_done;
};

// raise the dynamic-LDS limit of `kern` on the ctx's device once (hipFuncSetAttribute is per device, not per process)
static inline int dat_ensure_lds(dat_ctx* ctx, const void* kern, int bytes) {
    if (ctx->lds_attr_done.count(kern)) return DAT_OK;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        ctx->last_error = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
        return DAT_ERR_LAUNCH;
    }
    ctx->lds_attr_done.insert(kern);
    return DAT_OK;
}

#define DAT_FAIL(ctx, code, ...)                                  \
    do {                                                          \
        char _b[512];                                             \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                    \
        if (ctx) (ctx)->last_error = _b;                          \
        return (code);                                            \
    } while (0)

#define DAT_CHECK_LAUNCH(ctx, what)                                                         \
    do {                                                                                    \
        hipError_t _e = hipGetLastError();                                                  \
        if (_e != hipSuccess) DAT_FAIL(ctx, DAT_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(_e)); \
    } while (0)

#define DAT_ENFORCE(ctx, cond, ...)                               \
    do {                                                          \
        if (!(cond)) DAT_FAIL(ctx, DAT_ERR_ARG, __VA_ARGS__);     \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- the 16-bit element format of this BUILD --------------------------------------------------------------------------------------------
// libdat_hip.so: bfloat16 (the benched performance mode).  libdat_hip_f16.so (-DDAT_H16_IS_FP16, round 6): IEEE half -- the same enum value
// DAT_BF16 ("the 16-bit activation / weight type"), the same kernels, tiles and layouts; only the conversions below and the MFMA opcode
// differ (v_mfma_f32_32x32x16_f16 runs at the bf16 rate on gfx950).  Three more mantissa bits for activations that are O(1 - 1000) after
// the affine layers; the range shrinks to 65504 (values beyond saturate to +-inf).  One flavour per process (dat_h16_format()).
#ifdef DAT_H16_IS_FP16
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8_t;
typedef _Float16 dat_h16x2_t __attribute__((ext_vector_type(2)));
typedef float dat_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }       // round to nearest even
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
    const dat_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dat_h16x2_t));
}
__device__ __forceinline__ float bf2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
#define DAT_MFMA16(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8_t, A_), __builtin_bit_cast(h16x8_t, B_), C_, 0, 0, 0)
#define DAT_MFMA16_OP "v_mfma_f32_32x32x16_f16"
#define DAT_H16_FORMAT 1
#else
// fp32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32); the integer-arithmetic
// version it replaces cost ~8 VALU operations per element and dominated the epilogue of the memory-bound layers
typedef __bf16 dat_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float dat_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
// two values -> one packed dword (lo in bits 0-15)
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
    const dat_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dat_bf16x2_t));
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
#define DAT_MFMA16(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A_), __builtin_bit_cast(bf16x8_t, B_), C_, 0, 0, 0)
#define DAT_MFMA16_OP "v_mfma_f32_32x32x16_bf16"
#define DAT_H16_FORMAT 0
#endif

template <int DT> struct ElemOf;
template <> struct ElemOf<DAT_F32> {
    typedef float type;
    static constexpr int size = 4;
    __device__ static __forceinline__ float ld(const void* p, size_t i) { return ((const float*)p)[i]; }
    __device__ static __forceinline__ void st(void* p, size_t i, float v) { ((float*)p)[i] = v; }
};
template <> struct ElemOf<DAT_BF16> {
    typedef uint16_t type;
    static constexpr int size = 2;
    __device__ static __forceinline__ float ld(const void* p, size_t i) { return bf2f(((const uint16_t*)p)[i]); }
    __device__ static __forceinline__ void st(void* p, size_t i, float v) { ((uint16_t*)p)[i] = f2bf(v); }
};

// element `i` of a small by-value kernel-argument array, chosen with an unrolled compare-and-select: indexing such an array with a
// run-time `i` makes the compiler copy the whole argument struct to scratch memory first (det_select / det_limit_emit / nms_mask /
// rpn_select_sort_decode each carried 216-272 bytes of scratch per lane for it)
template <typename T, int N>
__device__ __forceinline__ T dat_pick(const T (&a)[N], int i) {
    T r = a[0];
#pragma unroll
    for (int k = 1; k < N; ++k)
        if (i == k) r = a[k];
    return r;
}

// grow the ctx-owned scratch (synchronises + reallocates only when it must grow); defined in c_api.hip
int dat_ensure_ws(dat_ctx* ctx, size_t bytes);

static inline int dat_esize(int dtype) { return dtype == DAT_BF16 ? 2 : 4; }
static inline long long cdiv_ll(long long a, long long b) { return (a + b - 1) / b; }
