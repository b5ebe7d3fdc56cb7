// Weight packing for the conv kernels: fp32 masters in the reference's blob layout [Cout, Cin, KT, KH, KW] -> the activation dtype in
// tap-major order, either row-major [tap][Cout_pad][Cin] (LDS-staged weight tiles) or MFMA A-fragment order (weights straight into
// registers); the data-gradient twin (channels swapped, taps flipped, AffineChannelNd scale folded in); the
// batched re-pack of a training step; and the first formulation of conv1 (dat_stem_pack / dat_stem_weights).
#include "conv_internal.h"

using namespace dat_conv;

namespace {

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 [Cout_real, Cin_real, KT, KH, KW] -> [tap][Cout_pad][Cin] in dtype, zero padded
// frag = 1: MFMA A-fragment order of the WD kernel variants: [tap][channel chunk of 128 B][32-row block][k-slice][lane][16 B],
// lane = k-half * 32 + row, the 16-B slot (2 * k-slice + k-half) of the row's 128-B chunk (what swz() addresses in the LDS path)
// dgrad = 1: pack the weights of the DATA-GRADIENT conv straight from the forward master w [CoutF = Cin_real][CinF = Cout_real][taps]:
// logical W'[co'][ci'][tap'] = w[ci'][co'][ntap - 1 - tap'] * scale[ci']  (channels swapped, every kernel axis flipped, the fused
// AffineChannelNd scale folded in) -- what the host used to build with flip + transpose + mul + contiguous before packing.
template <int DT>
__global__ void pack_weights_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout_real, int Cin_real,
                                    int ntap, int Cout_pad, int Cin, int frag, int dgrad, const float* __restrict__ scale) {
    constexpr int CK = Mma<DT>::CK, EPS = 16 / ElemOf<DT>::size;   // channels per 128-B chunk, elements per 16-B slot
    const size_t total = (size_t)ntap * Cout_pad * Cin;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = i % Cin;
        const int co = (i / Cin) % Cout_pad;
        const int tap = i / ((size_t)Cin * Cout_pad);
        float v = 0.f;
        if (co < Cout_real && ci < Cin_real) {
            if (dgrad) v = w[((size_t)ci * Cout_real + co) * ntap + (ntap - 1 - tap)] * (scale ? scale[ci] : 1.f);
            else v = w[((size_t)co * Cin_real + ci) * ntap + tap];
        }
        size_t dst = i;
        if (frag) {
            const int cc = ci / CK, cl = ci % CK, slot = cl / EPS, e = cl % EPS;
            const int lane = (slot & 1) * 32 + (co & 31);
            dst = (((((size_t)tap * (Cin / CK) + cc) * (Cout_pad >> 5) + (co >> 5)) * 4 + (slot >> 1)) * 64 + lane) * EPS + e;
        }
        ElemOf<DT>::st(out, dst, v);
    }
}

// The same packing through LDS: one block packs a tile of 32 output rows x 16 input channels for ALL taps.  The master layout has
// the taps innermost, the packed layout has them outermost, so the element-wise kernel above reads with a stride of ntap floats;
// training re-packs every trainable layer (and its data-gradient twin) after every SGD step, where that gather cost ~1.3 ms per
// iteration.  Here both sides are coalesced: the tile is read as contiguous runs (16 x ntap floats per row; in dgrad mode 32 x ntap
// floats per source row, the source being [CoutF][CinF][taps] with the dgrad's output channels second) and written as 16-byte
// pieces, 32 consecutive rows (= lanes of a fragment) per 512-byte run.
template <int DT>
__device__ __forceinline__ void pack_tile(const float* __restrict__ w, void* __restrict__ out, int Cout_real, int Cin_real, int ntap,
                                          int Cout_pad, int Cin, int frag, int dgrad, const float* __restrict__ scale, int co0, int ci0,
                                          int CIT) {
    // CIT: input channels of the tile (16 | 32 | 64, pack_cit): pointwise layers -- most of a bottleneck network's parameters -- get
    // 256-byte source runs and 8 KB per block instead of 64-byte runs and 2 KB
    constexpr int CT = 32;
    (void)CT;
    constexpr int CK = Mma<DT>::CK, EPS = 16 / ElemOf<DT>::size;
    extern __shared__ float tile[];                    // [CT co][CIT ci][ntap], rows padded by one float: the store phase reads with
                                                       // co across the lanes, and CIT * ntap (e.g. 432) is a multiple of 16 banks
    const int tid = threadIdx.x;
    const int per_co = CIT * ntap;
    const int pitch = per_co + 1;
    // 16-byte loads where the runs allow it (every conv layer but the stem: whole 16-channel / 32-row runs inside the tensor, 16-byte
    // aligned): a quarter of the load instructions and index divisions, four times the bytes in flight per wave -- the re-pack after an
    // SGD step runs at 2 blocks per CU (55 KB of LDS each) and was latency-bound at 0.4 TB/s
    const bool vec = ((size_t)w & 15) == 0 && ((size_t)(dgrad ? Cout_real : Cin_real) * ntap) % 4 == 0;   // (tile origins are multiples of 16 / 32)
    if (!dgrad && vec && ci0 + CIT <= Cin_real) {
        const int per4 = per_co >> 2;
        for (int L = tid; L < CT * per4; L += 256) {
            const int co_l = L / per4, e = (L - co_l * per4) << 2;
            const int co = co0 + co_l;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < Cout_real) v = *(const float4*)(w + ((size_t)co * Cin_real + ci0) * ntap + e);
            float* t = tile + co_l * pitch + e;
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        }
    } else if (dgrad && vec && co0 + CT <= Cout_real) {
        const int per_ci = CT * ntap, per4 = per_ci >> 2;
        for (int L = tid; L < CIT * per4; L += 256) {
            const int ci_l = L / per4, e0 = (L - ci_l * per4) << 2;
            const int ci = ci0 + ci_l;
            float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci < Cin_real) {
                v4 = *(const float4*)(w + ((size_t)ci * Cout_real + co0) * ntap + e0);
                if (scale) { const float sc = scale[ci]; v4.x *= sc; v4.y *= sc; v4.z *= sc; v4.w *= sc; }
            }
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
            int co_l = e0 / ntap, tsrc = e0 - co_l * ntap;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tile[co_l * pitch + ci_l * ntap + (ntap - 1 - tsrc)] = vv[k];
                if (++tsrc == ntap) { tsrc = 0; ++co_l; }
            }
        }
    } else if (!dgrad) {
        for (int L = tid; L < CT * per_co; L += 256) {
            const int co_l = L / per_co, e = L - co_l * per_co;
            const int co = co0 + co_l, ci = ci0 + e / ntap;
            tile[co_l * pitch + e] = (co < Cout_real && ci < Cin_real) ? w[((size_t)co * Cin_real + ci0) * ntap + e] : 0.f;
        }
    } else {
        const int per_ci = CT * ntap;
        for (int L = tid; L < CIT * per_ci; L += 256) {
            const int ci_l = L / per_ci, e = L - ci_l * per_ci;
            const int co_l = e / ntap, tsrc = e - co_l * ntap;
            const int co = co0 + co_l, ci = ci0 + ci_l;
            float v = 0.f;
            if (co < Cout_real && ci < Cin_real) v = w[((size_t)ci * Cout_real + co0) * ntap + e] * (scale ? scale[ci] : 1.f);
            tile[co_l * pitch + ci_l * ntap + (ntap - 1 - tsrc)] = v;
        }
    }
    __syncthreads();
    const int SL = CIT / EPS;                          // 16-byte pieces per row of the tile
    const int npieces = ntap * SL * CT;
    for (int id = tid; id < npieces; id += 256) {
        const int co_l = id % CT, r = id / CT, sl = r % SL, tap = r / SL;
        const int co = co0 + co_l, ci = ci0 + sl * EPS;
        float v[EPS];
#pragma unroll
        for (int e = 0; e < EPS; ++e) v[e] = tile[co_l * pitch + (sl * EPS + e) * ntap + tap];
        size_t dst;                                     // in elements
        if (frag) {
            const int cc = ci / CK, slot = (ci % CK) / EPS;
            const int lane = (slot & 1) * 32 + (co & 31);
            dst = (((((size_t)tap * (Cin / CK) + cc) * (Cout_pad >> 5) + (co >> 5)) * 4 + (slot >> 1)) * 64 + lane) * EPS;
        } else {
            dst = ((size_t)tap * Cout_pad + co) * Cin + ci;
        }
        uint4 o;
        if (DT == DAT_BF16) {
            o.x = f2bf2(v[0], v[1]); o.y = f2bf2(v[2], v[3]); o.z = f2bf2(v[4 % EPS], v[5 % EPS]); o.w = f2bf2(v[6 % EPS], v[7 % EPS]);
        } else {
            o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]); o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
        }
        *(uint4*)((char*)out + dst * ElemOf<DT>::size) = o;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void pack_weights_tiled_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout_real,
                                                                 int Cin_real, int ntap, int Cout_pad, int Cin, int frag, int dgrad,
                                                                 const float* __restrict__ scale) {
    pack_tile<DT>(w, out, Cout_real, Cin_real, ntap, Cout_pad, Cin, frag, dgrad, scale, blockIdx.x * 32, blockIdx.y * 16, 16);
}

// Batched re-pack (training): after an SGD step every trainable layer and its data-gradient twin is re-packed from the fp32 masters --
// ~100 launches of 3-20 us each, 1.75 ms of a 23 ms iteration.  One launch over a table of entries: block b belongs to the entry whose
// [tile0, tile0 + tiles) range holds it.  The entry is found with ONE round of loads -- every thread tests one entry, the block counts
// the entries that start at or before it -- where a binary search chained ~8 dependent L2 round trips in front of a block that moves 3 KB
// (round 6: the re-pack of R-50 ran at 0.34 TB/s, 0.44 ms per iteration, most of it this search and 2-KB tiles at 2 blocks per CU: the
// launches are now grouped by tap count on the host, so a pointwise entry no longer reserves the 55 KB of LDS a 27-tap tile needs).
template <int DT>
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const dat_pack_item* __restrict__ items, int n) {
    const int b = blockIdx.x;
    int cnt = 0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + (int)threadIdx.x;
        cnt += __syncthreads_count(i < n && items[i].tile0 <= b);
    }
    const dat_pack_item it = items[cnt - 1];          // (tile0 ascending from 0: cnt >= 1)
    const int local = b - it.tile0;
    const int bx = local % it.tiles_x, by = local / it.tiles_x;
    pack_tile<DT>(it.w, it.packed, it.rows, it.cols, it.ntap, it.cout_pad, it.cin, it.frag, it.dgrad, it.scale, bx * 32, by * it.cit, it.cit);
}

// input channels per tile of a batched entry: as many as keep the tile (32 rows x cit x taps floats) at or below ~37 KB
__host__ __device__ inline int pack_cit(int ntap) { return ntap <= 4 ? 64 : ntap <= 9 ? 32 : 16; }

// stem packing (see dat_hip.h: dat_stem_pack): one thread = one 16-byte group of output channels
template <int DT>
__global__ void stem_pack_kernel(const float* __restrict__ data, void* __restrict__ out, int N, int T, int H, int W,
                                 int Ho, int Wo) {
    constexpr int V = 16 / ElemOf<DT>::size;   // channels per thread
    constexpr int G = 64 / V;                  // groups per position
    const int R = Ho + 3;
    const size_t total = (size_t)N * T * R * Wo * G;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = i % G;
        size_t q = i / G;
        const int ow = q % Wo; q /= Wo;
        const int r = q % R; q /= R;
        const int t = q % T;
        const int n = q / T;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int ch = g * V + e;
            const int dkh = ch >> 5, rem = ch & 31;
            float x = 0.f;
            if (rem < 21) {
                const int kw = rem / 3, c = rem - kw * 3;
                const int ih = 2 * r - 3 + dkh, iw = 2 * ow - 3 + kw;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) x = data[((((size_t)n * 3 + c) * T + t) * H + ih) * W + iw];
            }
            v[e] = x;
        }
        uint4 o;
        if (DT == DAT_BF16) {
            o.x = f2bf2(v[0], v[1]); o.y = f2bf2(v[2], v[3]);
            o.z = f2bf2(v[4 % V], v[5 % V]); o.w = f2bf2(v[6 % V], v[7 % V]);
        } else {
            o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]); o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
        }
        *(uint4*)((char*)out + i * 16) = o;
    }
}

// conv1_w [Cout,3,1,7,7] -> [Cout,64,1,4,1]; channel dkh*32 + kw*3 + c of tap j holds w[co,c,0,2j+dkh,kw]
__global__ void stem_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout) {
    const int total = Cout * 64 * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 3;
        const int ch = (i >> 2) & 63;
        const int co = i >> 8;
        const int dkh = ch >> 5, rem = ch & 31;
        float v = 0.f;
        const int kh = 2 * j + dkh;
        if (rem < 21 && kh < 7) {
            const int kw = rem / 3, c = rem - kw * 3;
            v = w[(((size_t)co * 3 + c) * 7 + kh) * 7 + kw];
        }
        out[i] = v;
    }
}

// rows / cols: real extents of the packed matrix (forward: Cout, Cin; dgrad: CinF, CoutF)
int launch_pack(dat_ctx* ctx, hipStream_t st, const dat_conv_desc* d, const float* w, int rows, int cols, int dgrad, const float* scale,
                void* packed) {
    const int ntap = d->KT * d->KH * d->KW;
    const int cp = cout_pad_of(d);
    const int frag = weights_direct(ctx, d) ? 1 : 0;
    const size_t lds = (size_t)32 * (16 * ntap + 1) * sizeof(float);
    if (lds <= 160 * 1024 && !ctx->dbg_pack_simple) {     // coalesced on both sides (see pack_weights_tiled_kernel)
        const dim3 grid(cp / 32, d->Cin / 16);
        int rc;
        if (d->dtype == DAT_BF16) {
            if ((rc = dat_ensure_lds(ctx, (const void*)pack_weights_tiled_kernel<DAT_BF16>, 160 * 1024)) != DAT_OK) return rc;
            hipLaunchKernelGGL(pack_weights_tiled_kernel<DAT_BF16>, grid, dim3(256), lds, st, w, packed, rows, cols, ntap, cp, d->Cin, frag,
                               dgrad, scale);
        } else {
            if ((rc = dat_ensure_lds(ctx, (const void*)pack_weights_tiled_kernel<DAT_F32>, 160 * 1024)) != DAT_OK) return rc;
            hipLaunchKernelGGL(pack_weights_tiled_kernel<DAT_F32>, grid, dim3(256), lds, st, w, packed, rows, cols, ntap, cp, d->Cin, frag,
                               dgrad, scale);
        }
    } else {
        const size_t total = (size_t)ntap * cp * d->Cin;
        const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
        if (d->dtype == DAT_BF16)
            hipLaunchKernelGGL(pack_weights_kernel<DAT_BF16>, dim3(blocks), dim3(256), 0, st, w, packed, rows, cols, ntap, cp, d->Cin, frag,
                               dgrad, scale);
        else
            hipLaunchKernelGGL(pack_weights_kernel<DAT_F32>, dim3(blocks), dim3(256), 0, st, w, packed, rows, cols, ntap, cp, d->Cin, frag,
                               dgrad, scale);
    }
    DAT_CHECK_LAUNCH(ctx, "pack_weights");
    return DAT_OK;
}

}  // namespace

extern "C" {

int dat_conv3d_pack_weights(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const float* w, int Cout_real,
                            int Cin_real, void* packed) {
    DAT_ENFORCE(ctx, d && w && packed, "conv3d_pack_weights: null argument");
    DAT_ENFORCE(ctx, Cout_real <= d->Cout && Cin_real <= d->Cin, "conv3d_pack_weights: real dims exceed descriptor");
    return launch_pack(ctx, (hipStream_t)s, d, w, Cout_real, Cin_real, 0, nullptr, packed);
}

int dat_conv3d_pack_weights_dgrad(dat_ctx* ctx, dat_stream s, const dat_conv_desc* d, const float* w_fwd, int CoutF, int CinF,
                                  const float* scale_fwd, void* packed) {
    DAT_ENFORCE(ctx, d && w_fwd && packed, "conv3d_pack_weights_dgrad: null argument");
    DAT_ENFORCE(ctx, CinF <= d->Cout && CoutF <= d->Cin, "conv3d_pack_weights_dgrad: forward dims %d x %d exceed the data-gradient descriptor (%d outputs, %d inputs)",
                CoutF, CinF, d->Cout, d->Cin);
    return launch_pack(ctx, (hipStream_t)s, d, w_fwd, CinF, CoutF, 1, scale_fwd, packed);
}

int dat_conv3d_pack_item(dat_ctx* ctx, const dat_conv_desc* d, const float* w, int rows_real, int cols_real, int dgrad, const float* scale,
                         void* packed, dat_pack_item* item) {
    DAT_ENFORCE(ctx, d && w && packed && item, "conv3d_pack_item: null argument");
    DAT_ENFORCE(ctx, d->dtype == DAT_F32 || d->dtype == DAT_BF16, "conv3d_pack_item: bad dtype %d", d->dtype);
    if (dgrad) DAT_ENFORCE(ctx, rows_real <= d->Cout && cols_real <= d->Cin, "conv3d_pack_item: forward dims exceed the data-gradient descriptor");
    else DAT_ENFORCE(ctx, rows_real <= d->Cout && cols_real <= d->Cin, "conv3d_pack_item: real dims exceed descriptor");
    const int ntap = d->KT * d->KH * d->KW;
    const int cit = pack_cit(ntap);
    DAT_ENFORCE(ctx, (size_t)32 * (cit * ntap + 1) * sizeof(float) <= 128 * 1024, "conv3d_pack_item: %d taps exceed the LDS tile", ntap);
    DAT_ENFORCE(ctx, d->Cin % cit == 0, "conv3d_pack_item: Cin %d is not a multiple of the %d-channel tile", d->Cin, cit);
    item->w = w; item->packed = packed; item->scale = scale;
    item->rows = rows_real; item->cols = cols_real; item->ntap = ntap;
    item->cout_pad = cout_pad_of(d); item->cin = d->Cin;
    item->frag = weights_direct(ctx, d) ? 1 : 0;
    item->dgrad = dgrad ? 1 : 0; item->dtype = d->dtype;
    item->tile0 = 0; item->tiles_x = item->cout_pad / 32; item->cit = cit;
    return item->tiles_x * (d->Cin / cit);
}

int dat_conv3d_pack_weights_batch(dat_ctx* ctx, dat_stream s, const dat_pack_item* items_dev, int n, int total_blocks, int max_ntap,
                                  int dtype) {
    DAT_ENFORCE(ctx, items_dev && n > 0 && total_blocks > 0 && max_ntap > 0, "conv3d_pack_weights_batch: empty batch");
    DAT_ENFORCE(ctx, dtype == DAT_F32 || dtype == DAT_BF16, "conv3d_pack_weights_batch: bad dtype %d", dtype);
    size_t lds = 0;                                    // the largest tile an entry of <= max_ntap taps can have
    for (int t = 1; t <= max_ntap; ++t) lds = std::max(lds, (size_t)32 * (pack_cit(t) * t + 1) * sizeof(float));
    // (128 KiB: the block count of the entry search takes a few bytes of static LDS, so the full 160 KiB cannot be dynamic)
    DAT_ENFORCE(ctx, lds <= 128 * 1024, "conv3d_pack_weights_batch: %d taps exceed the LDS tile", max_ntap);
    int rc;
    if (dtype == DAT_BF16) {
        if ((rc = dat_ensure_lds(ctx, (const void*)pack_weights_batch_kernel<DAT_BF16>, 128 * 1024)) != DAT_OK) return rc;
        hipLaunchKernelGGL(pack_weights_batch_kernel<DAT_BF16>, dim3(total_blocks), dim3(256), lds, (hipStream_t)s, items_dev, n);
    } else {
        if ((rc = dat_ensure_lds(ctx, (const void*)pack_weights_batch_kernel<DAT_F32>, 128 * 1024)) != DAT_OK) return rc;
        hipLaunchKernelGGL(pack_weights_batch_kernel<DAT_F32>, dim3(total_blocks), dim3(256), lds, (hipStream_t)s, items_dev, n);
    }
    DAT_CHECK_LAUNCH(ctx, "pack_weights_batch");
    return DAT_OK;
}

int dat_stem_pack(dat_ctx* ctx, dat_stream s, const float* data, void* packed, int dtype, int N, int T, int H, int W) {
    DAT_ENFORCE(ctx, data && packed, "stem_pack: null argument");
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const size_t total = (size_t)N * T * (Ho + 3) * Wo * (64 / (16 / dat_esize(dtype)));
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 64);
    if (dtype == DAT_BF16)
        hipLaunchKernelGGL(stem_pack_kernel<DAT_BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)s, data, packed, N, T, H,
                           W, Ho, Wo);
    else
        hipLaunchKernelGGL(stem_pack_kernel<DAT_F32>, dim3(blocks), dim3(256), 0, (hipStream_t)s, data, packed, N, T, H, W,
                           Ho, Wo);
    DAT_CHECK_LAUNCH(ctx, "stem_pack");
    return DAT_OK;
}

int dat_stem_weights(dat_ctx* ctx, dat_stream s, const float* conv1_w, int Cout, float* w_k4) {
    DAT_ENFORCE(ctx, conv1_w && w_k4 && Cout > 0, "stem_weights: bad argument");
    hipLaunchKernelGGL(stem_weights_kernel, dim3((Cout * 256 + 255) / 256), dim3(256), 0, (hipStream_t)s, conv1_w, w_k4,
                       Cout);
    DAT_CHECK_LAUNCH(ctx, "stem_weights");
    return DAT_OK;
}

}  // extern "C"
