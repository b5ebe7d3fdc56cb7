// Training losses, forward value + gradient w.r.t. the prediction in one pass each (gfx950).
//
// Reference graph ops (Caffe2 / Detectron custom ops, semantics restated from their public definitions at the call
// sites lib/modeling/model_builder.py:481-494, 612-636, 873-889 and lib/modeling/FPN.py:282-321):
//   SigmoidCrossEntropyLoss(X, T int32, -1 = ignore)  loss = scale * sum_valid[max(x,0) - x*t + log(1+exp(-|x|))] / norm
//   SmoothL1Loss(Y, Yhat, a_in, a_out; beta)          v = a_in*(Y-Yhat); l = |v|<beta ? 0.5 v^2/beta : |v|-0.5 beta;
//                                                     loss = scale * sum(a_out * l) / N
//   SoftmaxWithLoss(X [R,D], label int32 [, w [R]])    loss = scale * sum_i -w_i log(max(p_i[label_i], 1e-20)) / norm
// The label tensors arrive from the host data loader in the reference's layouts; the "wide" RPN label arrays are
// narrowed to the head's H x W by indexing (SpatialNarrowAs, lib/ops/spatial_narrow_as_op.cu, folded away).
// Normalisers (number of non-ignored anchors, sum of weights, batch size) are passed in: the labels are host data.
#include "dat_common.h"

namespace {

__device__ __forceinline__ float ldf(const void* p, int dtype, size_t i) {
    return dtype == DAT_BF16 ? bf2f(((const uint16_t*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void stf(void* p, int dtype, size_t i, float v) {
    if (dtype == DAT_BF16) ((uint16_t*)p)[i] = f2bf(v);
    else ((float*)p)[i] = v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

struct RpnLossParams {
    const void* head;   // [N*Th, H, W, cs]: logits at logit_off + a, deltas at delta_off + ...
    void* dhead;        // same layout; every channel written (zeros outside logits/deltas)
    const int* labels;  // (N, A, Hw, Ww)
    const float* tgt;   // (N, 4*T*A, Hw, Ww), channel (a*T + t)*4 + c  (model_builder.py:545-563)
    const float* w_in;
    const float* w_out;
    int dtype, N, H, W, cs, A, logit_off, delta_off, Hw, Ww;
    int T;              // tube length of the anchors
    int per_frame;      // 1: the head keeps T frames per clip; logits are averaged over them, frame t holds the deltas
                        //    (a*4 + c) of tube slot t.  0: one frame per clip with 4*T*A delta channels
    float cls_mult;     // scale / norm
    float bbox_beta, bbox_mult;   // scale / N
    float* loss;        // [2]: cls, bbox (accumulated with atomics)
};

__global__ __launch_bounds__(256) void rpn_loss_kernel(const RpnLossParams p) {
    __shared__ float red[256];
    const int Th = p.per_frame ? p.T : 1;                  // frames per clip in the head tensor
    const int nd = p.per_frame ? 4 * p.A : 4 * p.A * p.T;  // delta channels per frame
    const long long npos = (long long)p.N * Th * p.H * p.W;
    const size_t fstride = (size_t)p.H * p.W * p.cs;
    float lc = 0.f, lb = 0.f;
    for (long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x; pos < npos; pos += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(pos % p.W);
        const int y = (int)((pos / p.W) % p.H);
        const int f = (int)(pos / ((long long)p.W * p.H));
        const int n = f / Th, t = f - n * Th;
        const size_t base = (size_t)pos * p.cs;
        for (int c = 0; c < p.cs; ++c) {
            const bool is_logit = c >= p.logit_off && c < p.logit_off + p.A;
            const bool is_delta = c >= p.delta_off && c < p.delta_off + nd;
            float grad = 0.f;
            if (is_logit) {
                const int a = c - p.logit_off;
                const int lab = p.labels[(((size_t)n * p.A + a) * p.Hw + y) * p.Ww + x];
                if (lab >= 0) {
                    float v = ldf(p.head, p.dtype, base + c);
                    if (p.per_frame) {      // TimePool 'avg' of the logits (model_builder.py:532): every frame gets d/T
                        v = 0.f;
                        const size_t b0 = base - (size_t)t * fstride;
                        for (int tt = 0; tt < Th; ++tt) v += ldf(p.head, p.dtype, b0 + (size_t)tt * fstride + c);
                        v /= (float)Th;
                    }
                    if (t == 0) lc += fmaxf(v, 0.f) - v * (float)lab + log1pf(expf(-fabsf(v)));
                    grad = (1.f / (1.f + expf(-v)) - (float)lab) * p.cls_mult / (float)Th;
                }
            } else if (is_delta) {
                const int ch = c - p.delta_off;
                const int lch = p.per_frame ? ((ch >> 2) * p.T + t) * 4 + (ch & 3) : ch;
                const size_t li = (((size_t)n * 4 * p.A * p.T + lch) * p.Hw + y) * p.Ww + x;
                const float wi = p.w_in[li], wo = p.w_out[li];
                const float v = wi * (ldf(p.head, p.dtype, base + c) - p.tgt[li]);
                const float av = fabsf(v);
                lb += wo * (av < p.bbox_beta ? 0.5f * v * v / p.bbox_beta : av - 0.5f * p.bbox_beta);
                const float dv = av < p.bbox_beta ? v / p.bbox_beta : (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));
                grad = wi * wo * dv * p.bbox_mult;
            }
            stf(p.dhead, p.dtype, base + c, grad);
        }
    }
    const float sc = block_sum(lc, red), sb = block_sum(lb, red);
    if (threadIdx.x == 0) {
        atomicAdd(p.loss + 0, sc * p.cls_mult);
        atomicAdd(p.loss + 1, sb * p.bbox_mult);
    }
}

__global__ __launch_bounds__(256) void smooth_l1_rows_kernel(const void* pred, int dtype, int ld, const float* tgt, const float* w_in,
                                                             const float* w_out, int R, int D, float beta, float mult, void* dpred,
                                                             float* loss) {
    __shared__ float red[256];
    float l = 0.f;
    const long long total = (long long)R * ld;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / ld), c = (int)(i - (long long)r * ld);
        float grad = 0.f;
        if (c < D) {
            const size_t li = (size_t)r * D + c;
            const float wi = w_in[li], wo = w_out[li];
            const float v = wi * (ldf(pred, dtype, i) - tgt[li]);
            const float av = fabsf(v);
            l += wo * (av < beta ? 0.5f * v * v / beta : av - 0.5f * beta);
            const float dv = av < beta ? v / beta : (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));
            grad = wi * wo * dv * mult;
        }
        stf(dpred, dtype, i, grad);
    }
    const float s = block_sum(l, red);
    if (threadIdx.x == 0) atomicAdd(loss, s * mult);
}

// one block per row; D up to a few thousand (56*56 spatial softmax)
__global__ __launch_bounds__(256) void softmax_ce_rows_kernel(const void* logits, int dtype, int ld, const int* labels,
                                                              const float* weights, int R, int D, float mult, void* dlogits,
                                                              int dl_dtype, int dl_ld, float* loss, int* correct) {
    __shared__ float red[256];
    __shared__ int redi[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)r * ld;
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int c = tid; c < D; c += blockDim.x) {
        const float v = ldf(logits, dtype, base + c);
        if (v > m) { m = v; am = c; }
    }
    red[tid] = m; redi[tid] = am;
    __syncthreads();
    for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
        if (tid < off) {
            if (red[tid + off] > red[tid] || (red[tid + off] == red[tid] && redi[tid + off] < redi[tid])) {
                red[tid] = red[tid + off]; redi[tid] = redi[tid + off];
            }
        }
        __syncthreads();
    }
    m = red[0];
    const int argmax = redi[0];
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < D; c += blockDim.x) s += expf(ldf(logits, dtype, base + c) - m);
    s = block_sum(s, red);
    const int label = labels[r];
    const float w = weights ? weights[r] : 1.f;
    for (int c = tid; c < dl_ld; c += blockDim.x) {
        float g = 0.f;
        if (c < D) {
            const float pr = expf(ldf(logits, dtype, base + c) - m) / s;
            g = w * (pr - (c == label ? 1.f : 0.f)) * mult;
        }
        stf(dlogits, dl_dtype, (size_t)r * dl_ld + c, g);
    }
    if (tid == 0) {
        const float pl = expf(ldf(logits, dtype, base + label) - m) / s;
        atomicAdd(loss, -w * logf(fmaxf(pl, 1e-20f)) * mult);
        if (correct && argmax == label) atomicAdd(correct, 1);
    }
}

}  // namespace

extern "C" {

int dat_rpn_loss(dat_ctx* ctx, dat_stream s, int dtype, const void* head, void* dhead, int N, int H, int W, int cstride, int A,
                 int T, int per_frame, int logit_off, int delta_off, const int* labels_wide, const float* targets_wide, const float* inside_wide,
                 const float* outside_wide, int Hw, int Ww, float cls_scale_over_norm, float bbox_beta,
                 float bbox_scale_over_n, float* loss2) {
    DAT_ENFORCE(ctx, head && dhead && labels_wide && targets_wide && inside_wide && outside_wide && loss2, "rpn_loss: null argument");
    DAT_ENFORCE(ctx, Hw >= H && Ww >= W, "rpn_loss: wide labels %dx%d smaller than the head %dx%d", Hw, Ww, H, W);
    DAT_ENFORCE(ctx, T >= 1 && logit_off + A <= cstride && delta_off + 4 * A * (per_frame ? 1 : T) <= cstride,
                "rpn_loss: head channels exceed the stride");
    RpnLossParams p;
    p.head = head; p.dhead = dhead; p.labels = labels_wide; p.tgt = targets_wide; p.w_in = inside_wide; p.w_out = outside_wide;
    p.dtype = dtype; p.N = N; p.H = H; p.W = W; p.cs = cstride; p.A = A; p.logit_off = logit_off; p.delta_off = delta_off;
    p.T = T; p.per_frame = per_frame;
    p.Hw = Hw; p.Ww = Ww; p.cls_mult = cls_scale_over_norm; p.bbox_beta = bbox_beta; p.bbox_mult = bbox_scale_over_n;
    p.loss = loss2;
    const long long npos = (long long)N * (per_frame ? T : 1) * H * W;
    long long blocks = (npos + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rpn_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p);
    DAT_CHECK_LAUNCH(ctx, "rpn_loss");
    return DAT_OK;
}

int dat_smooth_l1_rows(dat_ctx* ctx, dat_stream s, int dtype, const void* pred, int ld, const float* targets, const float* inside,
                       const float* outside, int R, int D, float beta, float scale_over_n, void* dpred, float* loss) {
    DAT_ENFORCE(ctx, pred && targets && inside && outside && dpred && loss && ld >= D, "smooth_l1_rows: bad argument");
    if (R == 0) return DAT_OK;
    long long blocks = ((long long)R * ld + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(smooth_l1_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, pred, dtype, ld, targets, inside,
                       outside, R, D, beta, scale_over_n, dpred, loss);
    DAT_CHECK_LAUNCH(ctx, "smooth_l1_rows");
    return DAT_OK;
}

int dat_softmax_ce_rows(dat_ctx* ctx, dat_stream s, int dtype, const void* logits, int ld, const int* labels, const float* weights,
                        int R, int D, float scale_over_norm, int dl_dtype, void* dlogits, int dl_ld, float* loss, int* correct) {
    DAT_ENFORCE(ctx, logits && labels && dlogits && loss && ld >= D && dl_ld >= D, "softmax_ce_rows: bad argument");
    if (R == 0) return DAT_OK;
    hipLaunchKernelGGL(softmax_ce_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)s, logits, dtype, ld, labels, weights, R, D,
                       scale_over_norm, dlogits, dl_dtype, dl_ld, loss, correct);
    DAT_CHECK_LAUNCH(ctx, "softmax_ce_rows");
    return DAT_OK;
}

}  // extern "C"
