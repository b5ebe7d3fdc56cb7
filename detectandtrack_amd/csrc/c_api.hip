// Context, error text and per-launch profiling of libdat_hip (see include/dat_hip.h).
#include <stdlib.h>

#include "dat_common.h"

int dat_ensure_ws(dat_ctx* ctx, size_t bytes) {
    if (ctx->ws_bytes >= bytes) return DAT_OK;
    // Growth is graph-safe: launches captured into a hipGraph have the OLD buffer's address baked in (proposal scratch, split-K
    // partials, bias partials) and may be replayed at any later time, so an outgrown buffer is retired, not freed -- it stays
    // valid until dat_ctx_destroy.  Buffers grow by >= 25 % a time, so the retired ones sum to less than 4x the live one.
    const size_t want = bytes + (bytes >> 2);
    void* grown = nullptr;
    if (hipMalloc(&grown, want) != hipSuccess) {
        (void)hipGetLastError();
        DAT_FAIL(ctx, DAT_ERR_ALLOC, "workspace hipMalloc(%zu) failed", want);
    }
    if (ctx->dbg_ws_poison && (hipMemset(grown, 0xFF, want) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) {   // (the fill runs on the null stream: wait for it)
        (void)hipGetLastError();
        hipFree(grown);
        DAT_FAIL(ctx, DAT_ERR_ALLOC, "workspace poison fill failed");
    }
    if (ctx->ws) ctx->ws_retired.push_back(ctx->ws);
    ctx->ws = grown;
    ctx->ws_bytes = want;
    ctx->ws_generation++;
    return DAT_OK;
}

extern "C" {

int dat_version(void) { return 1; }
int dat_h16_format(void) { return DAT_H16_FORMAT; }

int dat_ctx_create(dat_ctx** out, int device) {
    if (!out) return DAT_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return DAT_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return DAT_ERR_ARG;
    dat_ctx* c = new dat_ctx();
    c->device = device;
    c->prof_enabled = 0;
    c->prof_ev = nullptr;
    c->prof_cap = c->prof_n = 0;
    c->prof_flops = nullptr;
    c->prof_tag = nullptr;
    c->ws = nullptr;
    c->ws_bytes = 0;
    c->ws_generation = 0;
    c->zeros = nullptr;
    {
        auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        c->force_bp = env_int("DAT_CONV_BP", 0);
        c->force_ks = env_int("DAT_CONV_KSPLIT", 0);
        c->dbg_tw_log2 = env_int("DAT_CONV_TW_LOG2", -1);
        c->dbg_ablate = env_int("DAT_CONV_ABLATE", 0);
        c->dbg_lds_pad = env_int("DAT_CONV_LDS_PAD", 0);
        c->dbg_tps3 = env_int("DAT_CONV_TPS", 3) == 3;
        c->dbg_wd = env_int("DAT_CONV_WD", 2);       // 0 off, 1 the 128-channel tiles only, 2 (default) also the 64-channel ones
        c->dbg_ntap = env_int("DAT_CONV_NTAP", 1);   // 0 off, 1 default rule; bit 1 (2/3): unrolled 1x1 variant for every 1x1 layer; bit 2 (5): no dense stride-2 patches
        c->dbg_pack_simple = env_int("DAT_PACK_SIMPLE", 0) != 0;
        c->dbg_ws64 = env_int("DAT_CONV_WS64", 1);
        c->dbg_pwlw = env_int("DAT_CONV_PWLW", 1);
        c->dbg_pwks = env_int("DAT_CONV_PWKS", 8);
        c->dbg_wgrad_direct = env_int("DAT_WGRAD_DIRECT", 1);
        c->dbg_wgrad_ks = env_int("DAT_WGRAD_KS", 0);
        c->dbg_wgrad_dma = env_int("DAT_WGRAD_DMA", 1);
        c->dbg_ablate_wgrad = env_int("DAT_WGRAD_ABLATE", 0);
        c->dbg_wgrad_sub = env_int("DAT_WGRAD_SUB", 2);
        c->dbg_wgrad_ilv = env_int("DAT_WGRAD_ILV", 1);
        c->dbg_wgrad_xcd = env_int("DAT_WGRAD_XCD", 0);
        c->dbg_pw_xcd = env_int("DAT_CONV_PW_XCD", 1);
        c->dbg_wgrad_pw = env_int("DAT_WGRAD_PW", 1);
        c->dbg_kps_sep = env_int("DAT_KPS_DECODE_SEP", 1);
        c->dbg_linear = env_int("DAT_CONV_LINEAR", 1);
        c->dbg_order = env_int("DAT_CONV_ORDER", 0);
        c->dbg_persist_pct = env_int("DAT_PERSIST_PCT", 100);
        c->dbg_bt_min = env_int("DAT_CONV_BT_MIN", 390);   // smallest grid of the big-tile kernel, in hundredths of a round of the CUs
        c->dbg_bt = env_int("DAT_CONV_BT", 1);   // round 6: on for grids of >= 3.9 even rounds (FPN P2 output conv, conv_rpn_fpn2): +1.3 % on the R-18 forward, same box
        c->dbg_roi_fold = env_int("DAT_ROI_BWD_FOLD", 1);
        c->dbg_ws_poison = env_int("DAT_WS_POISON", 0);
        c->num_cu = 0;
    }
    // a private non-blocking stream + a pinned word buffer for the context's own small device <-> host transfers: the synchronous
    // hipMemcpy / hipMemset entry points are avoided after start-up (measured on ROCm 7.2: a synchronous device -> host hipMemcpy
    // between two replays of a captured hipGraph made the next replay fault)
    c->util_stream = nullptr;
    c->pinned = nullptr;
    if (hipStreamCreateWithFlags((hipStream_t*)&c->util_stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc(&c->pinned, 256, hipHostMallocDefault) != hipSuccess) {
        if (c->util_stream) hipStreamDestroy((hipStream_t)c->util_stream);
        delete c;
        return DAT_ERR_ALLOC;
    }
    // the all-zero line the conv kernels read for out-of-range patch pieces.  Zeroed on the context's own stream and WAITED for: a
    // plain hipMemset runs on the null stream, which the caller's non-blocking streams do not order against -- the first conv of a new
    // context (a pipeline slot's first forward, other forwards keeping the device busy) read the line before the fill had landed and
    // padded its border tiles with whatever the pages held (round 5: intermittent mismatch of the eager frame-trunk-cache pipeline)
    if (hipMalloc(&c->zeros, 512) != hipSuccess || hipMemsetAsync(c->zeros, 0, 512, (hipStream_t)c->util_stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)c->util_stream) != hipSuccess) {
        if (c->zeros) hipFree(c->zeros);
        hipHostFree(c->pinned);
        hipStreamDestroy((hipStream_t)c->util_stream);
        delete c;
        return DAT_ERR_ALLOC;
    }
    *out = c;
    return DAT_OK;
}

void dat_ctx_destroy(dat_ctx* ctx) {
    if (!ctx) return;
    if (ctx->prof_ev) {
        for (int i = 0; i < 2 * ctx->prof_cap; ++i) hipEventDestroy(ctx->prof_ev[i]);
        delete[] ctx->prof_ev;
        delete[] ctx->prof_flops;
        delete[] ctx->prof_tag;
    }
    if (ctx->ws) hipFree(ctx->ws);
    for (void* p : ctx->ws_retired) hipFree(p);
    if (ctx->zeros) hipFree(ctx->zeros);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    if (ctx->util_stream) hipStreamDestroy((hipStream_t)ctx->util_stream);
    delete ctx;
}

const char* dat_last_error(dat_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null dat_ctx"; }

int dat_ws_info(dat_ctx* ctx, void** ptr, size_t* bytes, int* generation) {
    if (!ctx) return DAT_ERR_ARG;
    if (ptr) *ptr = ctx->ws;
    if (bytes) *bytes = ctx->ws_bytes;
    if (generation) *generation = ctx->ws_generation;
    return DAT_OK;
}

int dat_ws_reserve(dat_ctx* ctx, size_t bytes) {
    if (!ctx) return DAT_ERR_ARG;
    return dat_ensure_ws(ctx, bytes);
}

int dat_prof_enable(dat_ctx* ctx, int capacity) {
    if (!ctx) return DAT_ERR_ARG;
    if (capacity <= 0) {
        ctx->prof_enabled = 0;
        return DAT_OK;
    }
    if (capacity > ctx->prof_cap) {
        if (ctx->prof_ev) {
            for (int i = 0; i < 2 * ctx->prof_cap; ++i) hipEventDestroy(ctx->prof_ev[i]);
            delete[] ctx->prof_ev;
            delete[] ctx->prof_flops;
            delete[] ctx->prof_tag;
        }
        ctx->prof_ev = new hipEvent_t[2 * capacity];
        for (int i = 0; i < 2 * capacity; ++i) hipEventCreate(&ctx->prof_ev[i]);
        ctx->prof_flops = new double[capacity];
        ctx->prof_tag = new int[capacity];
        ctx->prof_cap = capacity;
    }
    ctx->prof_n = 0;
    ctx->prof_enabled = 1;
    // clock counters (after everything in flight: the conv kernels of earlier launches add to them)
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemsetAsync((char*)ctx->zeros + 256, 0, 16, (hipStream_t)ctx->util_stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)ctx->util_stream) != hipSuccess) return DAT_ERR_LAUNCH;
    return DAT_OK;
}

int dat_prof_clock(dat_ctx* ctx, double* shader_mhz) {
    if (!ctx || !shader_mhz) return DAT_ERR_ARG;
    unsigned long long* h = (unsigned long long*)ctx->pinned;
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpyAsync(h, (char*)ctx->zeros + 256, 16, hipMemcpyDeviceToHost, (hipStream_t)ctx->util_stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)ctx->util_stream) != hipSuccess) return DAT_ERR_LAUNCH;
    *shader_mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;   // s_memrealtime ticks at 100 MHz
    return DAT_OK;
}

int dat_prof_read(dat_ctx* ctx, int max_records, int* tags, double* flops, float* ms) {
    if (!ctx) return DAT_ERR_ARG;
    const int n = ctx->prof_n < max_records ? ctx->prof_n : max_records;
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(ctx->prof_ev[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&t, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]) != hipSuccess) {
            (void)hipGetLastError();   // (never leave a sticky error behind for the caller's runtime to trip over)
            t = 0.f;
        }
        if (tags) tags[i] = ctx->prof_tag[i];
        if (flops) flops[i] = ctx->prof_flops[i];
        if (ms) ms[i] = t;
    }
    ctx->prof_n = 0;
    return n;
}

// zero `bytes` bytes at `ptr` on the stream (hipMemsetAsync; capturable): the zero-initialised outputs of the proposal / collect / detection
// kernels without a framework fill kernel in between
int dat_fill_zero(dat_ctx* ctx, dat_stream s, void* ptr, size_t bytes) {
    DAT_ENFORCE(ctx, ptr || bytes == 0, "fill_zero: null pointer");
    if (bytes == 0) return DAT_OK;
    if (hipMemsetAsync(ptr, 0, bytes, (hipStream_t)s) != hipSuccess) DAT_FAIL(ctx, DAT_ERR_LAUNCH, "fill_zero: hipMemsetAsync failed");
    return DAT_OK;
}

}  // extern "C"
