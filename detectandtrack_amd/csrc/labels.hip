// RPN anchor labelling, device half (training input pipeline).
//
// The reference labels every anchor of the field on the host (lib/roi_data/rpn.py:254-370): IoU of up to ~450 k anchors
// against the clip's ground-truth tubes with the Cython kernel (lib/utils/cython_bbox.pyx:16-57, averaged over the
// tube's frames by lib/utils/boxes.py:60-69), a row arg-max, a column max, and "every gt keeps its best anchors".
// That O(anchors x gts) part runs here, one thread per anchor, in the float/double evaluation order of the C that
// Cython emits so that the thresholded labels are identical; the host keeps only the two random sub-samplings.
// This file is compiled with -ffp-contract=off.
#include "dat_common.h"

namespace {

// IoU of one frame's anchor box against one gt box (cython_bbox.pyx:33-56): float differences, `+ 1` in double,
// the union formed in double and rounded once, float product and quotient.
__device__ __forceinline__ float box_iou(const float* __restrict__ b, const float* __restrict__ q) {
    const float iw = (float)((double)(fminf(b[2], q[2]) - fmaxf(b[0], q[0])) + 1.0);
    const float ih = (float)((double)(fminf(b[3], q[3]) - fmaxf(b[1], q[1])) + 1.0);
    if (!(iw > 0.f) || !(ih > 0.f)) return 0.f;
    const double bw = (double)(b[2] - b[0]) + 1.0, bh = (double)(b[3] - b[1]) + 1.0;
    const float qa = (float)(((double)(q[2] - q[0]) + 1.0) * ((double)(q[3] - q[1]) + 1.0));
    const float inter = iw * ih;
    const float ua = (float)(bw * bh + (double)qa - (double)inter);
    return inter / ua;
}

// mean over the tube's frames: float adds in frame order, one float division (np.mean over the per-frame matrices)
template <int MAXT>
__device__ __forceinline__ float tube_iou(const float* __restrict__ a, const float* __restrict__ q, int T) {
    float acc = box_iou(a, q);
    for (int t = 1; t < T; ++t) acc += box_iou(a + 4 * t, q + 4 * t);
    return T == 1 ? acc : acc / (float)T;
}

constexpr int LAB_MAXT = 8;     // frames per tube held in registers
constexpr int LAB_GT_LDS = 64;  // gts staged in LDS per pass

__device__ __forceinline__ bool anchor_inside(const float* a, int T, float lo, float xmax, float ymax) {
    bool ok = true;
    for (int t = 0; t < T; ++t)
        ok = ok && a[4 * t] >= lo && a[4 * t + 1] >= lo && a[4 * t + 2] < xmax && a[4 * t + 3] < ymax;
    return ok;
}

// pass 1: per anchor max / first arg-max over the gts (-1 / 0 for anchors that straddle the image border), and the
// per-gt maximum over the inside anchors (IoU >= 0, so the unsigned order of the float bits is the float order).
__global__ __launch_bounds__(256) void anchor_overlap_kernel(const float* __restrict__ anchors, int n, const float* __restrict__ gts, int G,
                                                             int T, int use_straddle, float lo, float xmax, float ymax,
                                                             float* __restrict__ a2g_max, int* __restrict__ a2g_arg,
                                                             unsigned int* __restrict__ gt_max) {
    __shared__ float sq[LAB_GT_LDS * 4 * LAB_MAXT];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a[4 * LAB_MAXT];
    bool inside = false;
    if (i < n) {
        for (int c = 0; c < 4 * T; ++c) a[c] = anchors[(size_t)i * 4 * T + c];
        inside = !use_straddle || anchor_inside(a, T, lo, xmax, ymax);
    }
    float best = inside && G > 0 ? -1.f : (inside ? 0.f : -1.f);
    int arg = 0;
    for (int g0 = 0; g0 < G; g0 += LAB_GT_LDS) {
        const int gn = min(LAB_GT_LDS, G - g0);
        __syncthreads();
        for (int e = threadIdx.x; e < gn * 4 * T; e += 256) sq[e] = gts[(size_t)g0 * 4 * T + e];
        __syncthreads();
        for (int g = 0; g < gn; ++g) {
            float v = inside ? tube_iou<LAB_MAXT>(a, sq + g * 4 * T, T) : 0.f;
            if (inside && v > best) { best = v; arg = g0 + g; }
            // wave-wide max, one atomic per wave and gt
            float m = v;
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(gt_max + g0 + g, __float_as_uint(m));
        }
    }
    if (i < n) { a2g_max[i] = best; a2g_arg[i] = arg; }
}

// pass 2: flag the anchors that attain some gt's maximum (rpn.py:300-305; a gt no inside anchor touches has maximum
// 0 and — as in the reference — flags every inside anchor with zero overlap)
__global__ __launch_bounds__(256) void anchor_best_kernel(const float* __restrict__ anchors, int n, const float* __restrict__ gts, int G, int T,
                                                          const float* __restrict__ a2g_max, const unsigned int* __restrict__ gt_max,
                                                          unsigned char* __restrict__ flag) {
    __shared__ float sq[LAB_GT_LDS * 4 * LAB_MAXT];
    __shared__ float sm[LAB_GT_LDS];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a[4 * LAB_MAXT];
    bool inside = false;
    if (i < n) {
        inside = a2g_max[i] >= 0.f;
        for (int c = 0; c < 4 * T; ++c) a[c] = anchors[(size_t)i * 4 * T + c];
    }
    bool hit = false;
    for (int g0 = 0; g0 < G; g0 += LAB_GT_LDS) {
        const int gn = min(LAB_GT_LDS, G - g0);
        __syncthreads();
        for (int e = threadIdx.x; e < gn * 4 * T; e += 256) sq[e] = gts[(size_t)g0 * 4 * T + e];
        for (int e = threadIdx.x; e < gn; e += 256) sm[e] = __uint_as_float(gt_max[g0 + e]);
        __syncthreads();
        if (inside)
            for (int g = 0; g < gn; ++g) hit = hit || tube_iou<LAB_MAXT>(a, sq + g * 4 * T, T) == sm[g];
    }
    if (i < n) flag[i] = hit ? 1 : 0;
}

__global__ void scatter_words_kernel(unsigned int* __restrict__ dst, long long dst_words, const int* __restrict__ offsets,
                                     const unsigned int* __restrict__ values, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const long long o = offsets[i];
        if (o >= 0 && o < dst_words) dst[o] = values[i];
    }
}

// ---- GenerateProposalLabels on the device (round 5) -------------------------------------------------------------
// lib/ops/generate_proposal_labels.py:24-37 -> roi_data/fast_rcnn.py:109-203 (+ json_dataset.py:423-473 merge, keypoint_rcnn.py:32-99,
// utils/keypoints.py:152-207): the clip's proposals are merged behind its ground-truth boxes, every candidate gets its maximum overlap
// with the gt boxes, BATCH_SIZE_PER_IM rois are drawn -- up to FG_FRACTION of them foreground (overlap >= FG_THRESH), the rest background
// ([BG_THRESH_LO, BG_THRESH_HI)) -- and turned into class labels, class-specific box targets and weights; the keypoint branch draws up to
// the same number of foreground rois that see a visible keypoint of their gt and builds the heatmap cell labels.
//
// What cannot be kept is NumPy's Mersenne-Twister stream (`npr.choice`): the device draws with a COUNTER-BASED generator instead.
// Contract (tests/test_gpu_train.py): the candidate sets (fg / bg / keypoint-fg) and the counts (n_fg, n_bg, n_kp) are the reference's;
// a draw of n out of a set S is "the n members of S with the smallest (key, index)", key = roi_key(seed, iteration, stream, index) below
// -- a uniformly random subset in uniformly random order when the keys are i.i.d. uniform, the order being the output row order;
// labels, targets, weights and heatmap cells of a drawn roi are the reference's arithmetic (float32, its operation order).
// Blocks of 1024 threads (<= 4096 candidates): every block computes overlaps, flags and keys of ALL candidates into its LDS; block b then
// ranks candidates [64 b, 64 b + 64) by counting (16 lanes share a candidate's scan over the N keys) and writes their rows.
__device__ __forceinline__ unsigned roi_key(unsigned seed_lo, unsigned seed_hi, unsigned iter, unsigned stream, unsigned index) {
    unsigned h = seed_lo ^ (index * 0x9E3779B9u) ^ (iter * 0x85EBCA6Bu) ^ (stream * 0xC2B2AE35u);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;      // murmur3 finaliser
    h ^= seed_hi;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

struct RoiSampleParams {
    dat_roi_sample_desc d;
    const float* props; const int* n_props; int props_cap;
    const float* gt_boxes; const int* gt_classes; const int* gt_kps; int G;
    float* rois; int* labels; float* targets; float* w_in; float* w_out;
    float* kp_rois; int* kp_loc; float* kp_w;
    int* counts; int* picked;
};

constexpr int RS_MAXN = 4096;

__global__ __launch_bounds__(1024) void roi_sample_kernel(const RoiSampleParams p) {
    __shared__ float s_mo[RS_MAXN];
    __shared__ unsigned s_key[RS_MAXN], s_kkey[RS_MAXN];
    __shared__ short s_arg[RS_MAXN];
    __shared__ unsigned char s_flag[RS_MAXN];       // 1 fg, 2 bg, 4 keypoint-fg
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x;
    const int T = p.d.T, C4 = 4 * T, G = p.G;
    const int P = min(*p.n_props, p.props_cap);
    const int N = G + P;
    const float scale = p.d.im_scale;
    const int KK = p.d.num_keypoints * T;           // keypoints of a tube
    if (tid < 4) s_cnt[tid] = 0;
    __syncthreads();
    // ---- per candidate: box at the image's own scale, max overlap / first arg-max over the gts, flags, keys ----
    for (int i = tid; i < N; i += 1024) {
        float box[4 * LAB_MAXT];
        float mo;
        int arg;
        if (i < G) {
            for (int c = 0; c < C4; ++c) box[c] = p.gt_boxes[(size_t)i * C4 + c];
            mo = 1.f; arg = i;                      // a gt row: overlap 1 with its own class (json_dataset.py: gt_overlaps one-hot)
        } else {
            for (int c = 0; c < C4; ++c) box[c] = p.props[(size_t)(i - G) * (C4 + 1) + 1 + c] / scale;
            mo = 0.f; arg = -1;
            for (int g = 0; g < G; ++g) {
                const float v = tube_iou<LAB_MAXT>(box, p.gt_boxes + (size_t)g * C4, T);
                if (v > mo) { mo = v; arg = g; }    // first maximum; rows whose maximum is 0 stay unassigned (fast_rcnn merge: `mx > 0`)
            }
        }
        unsigned char fl = 0;
        if (mo >= p.d.fg_thresh) fl |= 1;
        if (mo < p.d.bg_thresh_hi && mo >= p.d.bg_thresh_lo) fl |= 2;
        if ((fl & 1) && p.gt_kps && arg >= 0) {     // keypoint_rcnn.py:88-99: a visible keypoint of the roi's gt inside the roi's FIRST-frame box
            const int* kp = p.gt_kps + (size_t)arg * 3 * KK;
            bool vis = false;
            for (int k = 0; k < KK; ++k) {
                const float x = (float)kp[k], y = (float)kp[KK + k];
                vis = vis || (kp[2 * KK + k] > 0 && x >= box[0] && x <= box[2] && y >= box[1] && y <= box[3]);
            }
            if (vis) fl |= 4;
        }
        s_mo[i] = mo; s_arg[i] = (short)arg; s_flag[i] = fl;
        s_key[i] = roi_key(p.d.seed_lo, p.d.seed_hi, p.d.iter, 0u, (unsigned)i);
        s_kkey[i] = roi_key(p.d.seed_lo, p.d.seed_hi, p.d.iter, 1u, (unsigned)i);
        if (fl & 1) atomicAdd(&s_cnt[0], 1);
        if (fl & 2) atomicAdd(&s_cnt[1], 1);
        if (fl & 4) atomicAdd(&s_cnt[2], 1);
    }
    __syncthreads();
    const int all_fg = s_cnt[0], all_bg = s_cnt[1], all_kp = s_cnt[2];
    const int n_fg = min(p.d.fg_rois_per_im, all_fg);
    const int n_bg = min(p.d.rois_per_im - n_fg, all_bg);
    const bool kp_fallback = p.gt_kps && all_kp == 0;                   // keypoint_rcnn.py:47-48: no keypoint-fg roi -> the gt boxes themselves
    const int n_kp = !p.gt_kps ? 0 : kp_fallback ? min(G, p.d.fg_rois_per_im) : min(p.d.fg_rois_per_im, all_kp);
    if (tid == 0 && blockIdx.x == 0) {
        p.counts[0] = n_fg + n_bg; p.counts[1] = n_fg; p.counts[2] = n_kp; p.counts[3] = all_fg; p.counts[4] = all_bg; p.counts[5] = all_kp;
    }
    const int Kc = p.d.cls_agnostic ? 2 : p.d.num_classes;
    const int ld_t = C4 * Kc;
    const int M = p.d.heatmap_size;
    // ---- ranks and rows: block b owns candidates [64 b, 64 b + 64), sixteen lanes per candidate share the scan over all N keys ----
    // (every block has computed the same overlaps / flags / keys above: ~N / 512 candidates per thread, nothing next to the N x N
    //  comparisons below -- 414 us in one block, the first version; the grid spreads them over N / 64 CUs)
    {
        const int i = blockIdx.x * 64 + (tid >> 4), sub = tid & 15;
        const bool live = i < N;
        const unsigned char fl = live ? s_flag[i] : 0;
        // rank of this candidate inside each set it belongs to: members with a smaller (key, index)
        int r_fg = 0, r_bg = 0, r_kp = 0;
        if (fl & 3) {
            const unsigned ki = s_key[i];
            for (int j = sub; j < N; j += 16) {
                const unsigned kj = s_key[j];
                const bool before = kj < ki || (kj == ki && j < i);
                const unsigned char fj = s_flag[j];
                r_fg += (before && (fj & 1)) ? 1 : 0;
                r_bg += (before && (fj & 2)) ? 1 : 0;
            }
        }
        if (fl & 4) {
            const unsigned ki = s_kkey[i];
            for (int j = sub; j < N; j += 16) {
                const unsigned kj = s_kkey[j];
                r_kp += ((kj < ki || (kj == ki && j < i)) && (s_flag[j] & 4)) ? 1 : 0;
            }
        }
        for (int off = 8; off > 0; off >>= 1) {
            r_fg += __shfl_xor(r_fg, off);
            r_bg += __shfl_xor(r_bg, off);
            r_kp += __shfl_xor(r_kp, off);
        }
        int kvalid = 0;
        if (live && sub == 0) do {
        int row = -1;
        if ((fl & 1) && r_fg < n_fg) row = r_fg;
        else if ((fl & 2) && !((fl & 1) && r_fg < n_fg) && r_bg < n_bg) row = n_fg + r_bg;
        int krow = -1;
        if (kp_fallback) { if (i < G && i < n_kp) krow = i; }
        else if ((fl & 4) && r_kp < n_kp) krow = r_kp;
        if (row < 0 && krow < 0) break;
        float box[4 * LAB_MAXT];
        if (i < G) for (int c = 0; c < C4; ++c) box[c] = p.gt_boxes[(size_t)i * C4 + c];
        else for (int c = 0; c < C4; ++c) box[c] = p.props[(size_t)(i - G) * (C4 + 1) + 1 + c] / scale;
        const int arg = s_arg[i];
        if (row >= 0) {
            if (p.picked) p.picked[row] = i;
            float* ro = p.rois + (size_t)row * (C4 + 1);
            ro[0] = 0.f;
            for (int c = 0; c < C4; ++c) ro[1 + c] = box[c] * scale;
            const int cls = row < n_fg ? (arg >= 0 ? p.gt_classes[arg] : 0) : 0;           // fast_rcnn.py:156-158: background rows get label 0
            p.labels[row] = cls;
            float* tg = p.targets + (size_t)row * ld_t;
            float* wi = p.w_in + (size_t)row * ld_t;
            float* wo = p.w_out + (size_t)row * ld_t;
            for (int c = 0; c < ld_t; ++c) { tg[c] = 0.f; wi[c] = 0.f; wo[c] = 0.f; }
            if (cls > 0 && arg >= 0) {
                const float* gb = p.gt_boxes + (size_t)arg * C4;
                const int slot = (p.d.cls_agnostic ? 1 : cls) * C4;
                for (int t = 0; t < T; ++t) {
                    // utils/boxes.bbox_transform_inv (:205-239), float32, its operation order
                    const float* e = box + 4 * t;
                    const float* q = gb + 4 * t;
                    const float ew = e[2] - e[0] + 1.0f, eh = e[3] - e[1] + 1.0f, gw = q[2] - q[0] + 1.0f, gh = q[3] - q[1] + 1.0f;
                    tg[slot + 4 * t + 0] = p.d.reg_weights[0] * ((q[0] + 0.5f * gw) - (e[0] + 0.5f * ew)) / ew;
                    tg[slot + 4 * t + 1] = p.d.reg_weights[1] * ((q[1] + 0.5f * gh) - (e[1] + 0.5f * eh)) / eh;
                    tg[slot + 4 * t + 2] = p.d.reg_weights[2] * logf(gw / ew);
                    tg[slot + 4 * t + 3] = p.d.reg_weights[3] * logf(gh / eh);
                    for (int c = 0; c < 4; ++c) { wi[slot + 4 * t + c] = 1.f; wo[slot + 4 * t + c] = 1.f; }
                }
            }
        }
        if (krow >= 0) {
            if (p.picked) p.picked[p.d.rois_per_im + krow] = i;
            float* ro = p.kp_rois + (size_t)krow * (C4 + 1);
            ro[0] = 0.f;
            for (int c = 0; c < C4; ++c) ro[1 + c] = box[c] * scale;
            int* lo = p.kp_loc + (size_t)krow * KK;
            float* wt = p.kp_w + (size_t)krow * KK;
            const int Kf = p.d.num_keypoints;
            for (int t = 0; t < T; ++t) {
                // utils/keypoints.py:152-207 keypoints_to_heatmap_labels on frame t's box and keypoints
                const float x1 = box[4 * t], y1 = box[4 * t + 1], x2 = box[4 * t + 2], y2 = box[4 * t + 3];
                const float sx = (float)M / (x2 - x1 + 1.0f), sy = (float)M / (y2 - y1 + 1.0f);
                for (int k = 0; k < Kf; ++k) {
                    int cell = 0;
                    float w = 0.f;
                    if (arg >= 0) {
                        const int* kp = p.gt_kps + (size_t)arg * 3 * KK;
                        const float xf = (float)kp[t * Kf + k], yf = (float)kp[KK + t * Kf + k];
                        const bool vis = kp[2 * KK + t * Kf + k] > 0;
                        float x = floorf((xf - x1) * sx), y = floorf((yf - y1) * sy);
                        if (xf == x2) x = (float)(M - 1);
                        if (yf == y2) y = (float)(M - 1);
                        if (x >= 0.f && y >= 0.f && x < (float)M && y < (float)M && vis) { cell = (int)(y * (float)M + x); w = 1.f; }
                    }
                    lo[t * Kf + k] = cell;
                    wt[t * Kf + k] = w;
                    kvalid += w > 0.f ? 1 : 0;
                }
            }
        }
        } while (false);
        // labelled keypoints of the drawn keypoint rois (the loss normaliser, model_builder.py:873-905): counts[6] is zeroed by the caller
        if (kvalid) atomicAdd(p.counts + 6, kvalid);
    }
}

}  // namespace

extern "C" {

int dat_anchor_overlaps(dat_ctx* ctx, dat_stream s, const float* anchors, int n, const float* gts, int G, int T, float im_h, float im_w,
                        float straddle, float* a2g_max, int* a2g_arg, unsigned char* best_flag, unsigned int* gt_max) {
    DAT_ENFORCE(ctx, anchors && a2g_max && a2g_arg && best_flag && gt_max, "anchor_overlaps: null argument");
    DAT_ENFORCE(ctx, T >= 1 && T <= LAB_MAXT, "anchor_overlaps: T %d must be 1..%d", T, LAB_MAXT);
    DAT_ENFORCE(ctx, G == 0 || gts, "anchor_overlaps: null gts");
    if (n == 0) return DAT_OK;
    hipStream_t st = (hipStream_t)s;
    if (G > 0 && hipMemsetAsync(gt_max, 0, sizeof(unsigned int) * G, st) != hipSuccess) DAT_FAIL(ctx, DAT_ERR_LAUNCH, "anchor_overlaps: memset failed");
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const int use_straddle = straddle >= 0.f;
    hipLaunchKernelGGL(anchor_overlap_kernel, dim3(blocks), dim3(256), 0, st, anchors, n, gts, G, T, use_straddle, -straddle, im_w + straddle,
                       im_h + straddle, a2g_max, a2g_arg, gt_max);
    DAT_CHECK_LAUNCH(ctx, "anchor_overlaps");
    if (G > 0) {
        hipLaunchKernelGGL(anchor_best_kernel, dim3(blocks), dim3(256), 0, st, anchors, n, gts, G, T, a2g_max, gt_max, best_flag);
        DAT_CHECK_LAUNCH(ctx, "anchor_best");
    } else if (hipMemsetAsync(best_flag, 0, (size_t)n, st) != hipSuccess) {
        DAT_FAIL(ctx, DAT_ERR_LAUNCH, "anchor_overlaps: memset failed");
    }
    return DAT_OK;
}

int dat_scatter_words(dat_ctx* ctx, dat_stream s, void* dst, long long dst_words, const int* offsets, const void* values, int n) {
    DAT_ENFORCE(ctx, dst && (n == 0 || (offsets && values)), "scatter_words: null argument");
    if (n == 0) return DAT_OK;
    hipLaunchKernelGGL(scatter_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, (unsigned int*)dst, dst_words, offsets,
                       (const unsigned int*)values, n);
    DAT_CHECK_LAUNCH(ctx, "scatter_words");
    return DAT_OK;
}

int dat_sample_rois(dat_ctx* ctx, dat_stream s, const dat_roi_sample_desc* d, const float* props, const int* n_props, int props_cap,
                    const float* gt_boxes, const int* gt_classes, const int* gt_kps, int G, float* rois, int* labels, float* targets,
                    float* w_in, float* w_out, float* kp_rois, int* kp_loc, float* kp_w, int* counts, int* picked) {
    DAT_ENFORCE(ctx, d && props && n_props && gt_boxes && gt_classes && rois && labels && targets && w_in && w_out && counts,
                "sample_rois: null argument");
    DAT_ENFORCE(ctx, d->T >= 1 && d->T <= LAB_MAXT && G >= 1 && props_cap >= 0 && G + props_cap <= RS_MAXN,
                "sample_rois: T %d (1..%d), %d gts + %d proposals (<= %d candidates)", d->T, LAB_MAXT, G, props_cap, RS_MAXN);
    DAT_ENFORCE(ctx, d->rois_per_im >= 1 && d->fg_rois_per_im >= 0 && d->fg_rois_per_im <= d->rois_per_im && d->num_classes >= 2,
                "sample_rois: rois per image %d / %d foreground, %d classes", d->rois_per_im, d->fg_rois_per_im, d->num_classes);
    DAT_ENFORCE(ctx, !gt_kps || (kp_rois && kp_loc && kp_w && d->num_keypoints >= 1 && d->heatmap_size >= 1 && G <= d->fg_rois_per_im),
                "sample_rois: keypoint outputs missing (or more gts than foreground rois per image)");
    RoiSampleParams p;
    p.d = *d;
    p.props = props; p.n_props = n_props; p.props_cap = props_cap;
    p.gt_boxes = gt_boxes; p.gt_classes = gt_classes; p.gt_kps = gt_kps; p.G = G;
    p.rois = rois; p.labels = labels; p.targets = targets; p.w_in = w_in; p.w_out = w_out;
    p.kp_rois = kp_rois; p.kp_loc = kp_loc; p.kp_w = kp_w; p.counts = counts; p.picked = picked;
    DAT_ENFORCE(ctx, hipMemsetAsync(counts + 6, 0, 2 * sizeof(int), (hipStream_t)s) == hipSuccess, "sample_rois: memset failed");
    hipLaunchKernelGGL(roi_sample_kernel, dim3((unsigned)((G + props_cap + 63) / 64)), dim3(1024), 0, (hipStream_t)s, p);
    DAT_CHECK_LAUNCH(ctx, "sample_rois");
    return DAT_OK;
}

}  // extern "C"
