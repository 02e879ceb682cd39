// Box-regression loss of the detection head on its raw NCHW output (runs twice per iteration:
// student features and teacher features)  [ref: distillator.py:107-112 -> student.losses -> detectron2 RetinaNet.losses:
//  smooth_l1(pred_deltas[pos], Box2BoxTransform.get_deltas(anchors, matched_gt)[pos], beta, "sum")].
// The torch restatement materialises the target deltas for ALL 8 x 201,600 anchors, permutes / concatenates the predicted
// deltas to (N, R, 4) and runs ~12 elementwise kernels; only ~200 anchors per image are positive.  Here one kernel walks
// the int32 label planes (shared with the focal loss), and only for positives computes the target deltas from the anchor
// and its matched box and reads the 4 predicted deltas in place: element (n, a*4 + k, y, x) is anchor (y, x, a), coord k.
//   forward : sum over positives and coordinates (fp64 block partials, fixed-order reduction)
//   backward: d deltas in the same NCHW layout (dense: zeros for non-positives), scaled by the upstream gradient
#include "common.h"

namespace lgd {

struct BoxRegArgs {
    const float* d[LGD_MAX_LEVELS];        // (N, A*4, H, W)
    const int32_t* lab[LGD_MAX_LEVELS];    // (N, A, H, W); positives: 0 <= label < K
    float* gd[LGD_MAX_LEVELS];
    const float* anchors;                  // (R, 4), R ordered (level, y, x, a)
    const float* matched;                  // (N, R, 4)
    int HW[LGD_MAX_LEVELS], r0[LGD_MAX_LEVELS], blk0[LGD_MAX_LEVELS + 1];
    int L, N, A, K, R, nblk;
    float beta, wx, wy, ww, wh;
    double* ws;                            // [nblk] block partials
    float* loss;
    const float* gscale;
};

template <int MODE>  // 0 forward, 1 backward
__global__ __launch_bounds__(256) void box_reg_kernel(BoxRegArgs a) {
    int l = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) l += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    const int HW = a.HW[l];
    const long long i = (long long)(blockIdx.x - a.blk0[l]) * 256 + threadIdx.x;   // over N * A * HW
    const bool on = i < (long long)a.N * a.A * HW;
    double acc = 0.0;
    if (on) {
        const int e = (int)(i % HW), na = (int)(i / HW), an = na % a.A, n = na / a.A;
        const int lab = a.lab[l][i];
        const bool pos = lab >= 0 && lab < a.K;
        const size_t base = ((size_t)na * 4) * HW + e;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        if (pos) {
            const int r = a.r0[l] + e * a.A + an;
            const float4 s = reinterpret_cast<const float4*>(a.anchors)[r];
            const float4 t = reinterpret_cast<const float4*>(a.matched)[(size_t)n * a.R + r];
            // Box2BoxTransform.get_deltas
            const float sw = s.z - s.x, sh = s.w - s.y, sx = s.x + 0.5f * sw, sy = s.y + 0.5f * sh;
            const float dw = t.z - t.x, dh = t.w - t.y, dx = t.x + 0.5f * dw, dy = t.y + 0.5f * dh;
            const float tgt[4] = {a.wx * (dx - sx) / sw, a.wy * (dy - sy) / sh, a.ww * logf(dw / sw), a.wh * logf(dh / sh)};
            const float gs = MODE == 1 ? a.gscale[0] : 0.f;
            float part = 0.f;
            #pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float df = a.d[l][base + (size_t)k * HW] - tgt[k], ad = fabsf(df);
                if (MODE == 0) {
                    part += (a.beta >= 1e-5f && ad < a.beta) ? 0.5f * ad * ad / a.beta : (a.beta >= 1e-5f ? ad - 0.5f * a.beta : ad);
                } else {
                    const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
                    g[k] = gs * ((a.beta >= 1e-5f && ad < a.beta) ? df / a.beta : sg);
                }
            }
            acc = (double)part;
        }
        if (MODE == 1) {
            #pragma unroll
            for (int k = 0; k < 4; ++k) a.gd[l][base + (size_t)k * HW] = g[k];
        }
    }
    if (MODE == 0) {
        __shared__ double red[4];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) a.ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

__global__ __launch_bounds__(256) void box_reg_reduce_kernel(BoxRegArgs a) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < a.nblk; i += 256) s += a.ws[i];   // fixed assignment of partials to threads
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.loss[0] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

static int box_reg_fill(BoxRegArgs& a, const float* const* d, const int32_t* const* lab, const int32_t* level_hw, int L, int N,
                        int A, int K, const float* anchors, const float* matched, int R, float beta, const float* w4) {
    if (!d || !lab || !level_hw || !anchors || !matched || !w4 || L < 1 || L > LGD_MAX_LEVELS || N < 1 || A < 1) return LGD_EINVAL;
    a.L = L; a.N = N; a.A = A; a.K = K; a.R = R; a.anchors = anchors; a.matched = matched; a.beta = beta;
    a.wx = w4[0]; a.wy = w4[1]; a.ww = w4[2]; a.wh = w4[3];
    int r0 = 0, blk = 0;
    for (int l = 0; l < L; ++l) {
        if (!d[l] || !lab[l]) return LGD_EINVAL;
        a.d[l] = d[l]; a.lab[l] = lab[l]; a.gd[l] = nullptr;
        a.HW[l] = level_hw[2 * l] * level_hw[2 * l + 1];
        a.r0[l] = r0; a.blk0[l] = blk;
        r0 += a.HW[l] * A;
        blk += (int)(((long long)N * A * a.HW[l] + 255) / 256);
    }
    if (r0 != R) return LGD_EINVAL;
    a.blk0[L] = blk; a.nblk = blk;
    return LGD_OK;
}

}  // namespace lgd

extern "C" {

size_t lgd_box_reg_ws_doubles(const int32_t* level_hw_host, int L, int N, int A) {
    size_t blk = 0;
    for (int l = 0; l < L; ++l) blk += ((size_t)N * A * level_hw_host[2 * l] * level_hw_host[2 * l + 1] + 255) / 256;
    return blk;
}

int lgd_box_reg_loss_fwd(const float* const* deltas_host, const int32_t* const* labels_host, const int32_t* level_hw_host, int L,
                         int N, int A, int K, const float* anchors, const float* matched_boxes, int R, float beta,
                         const float* weights4_host, double* ws, float* loss, void* stream) {
    lgd::BoxRegArgs a;
    if (!ws || !loss || lgd::box_reg_fill(a, deltas_host, labels_host, level_hw_host, L, N, A, K, anchors, matched_boxes, R, beta,
                                          weights4_host) != LGD_OK) return LGD_EINVAL;
    a.ws = ws; a.loss = loss; a.gscale = nullptr;
    LGD_LAUNCH("box_reg_fwd_kernel", lgd::box_reg_kernel<0>, dim3(a.nblk), dim3(256), 0, (hipStream_t)stream, a);
    LGD_LAUNCH("box_reg_reduce_kernel", lgd::box_reg_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

int lgd_box_reg_loss_bwd(const float* const* deltas_host, const int32_t* const* labels_host, const int32_t* level_hw_host, int L,
                         int N, int A, int K, const float* anchors, const float* matched_boxes, int R, float beta,
                         const float* weights4_host, const float* grad_loss, float* const* grad_deltas_host, void* stream) {
    lgd::BoxRegArgs a;
    if (!grad_loss || !grad_deltas_host || lgd::box_reg_fill(a, deltas_host, labels_host, level_hw_host, L, N, A, K, anchors,
                                                             matched_boxes, R, beta, weights4_host) != LGD_OK) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!grad_deltas_host[l]) return LGD_EINVAL;
        a.gd[l] = grad_deltas_host[l];
    }
    a.ws = nullptr; a.loss = nullptr; a.gscale = grad_loss;
    LGD_LAUNCH("box_reg_bwd_kernel", lgd::box_reg_kernel<1>, dim3(a.nblk), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
