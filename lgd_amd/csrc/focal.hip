// Sigmoid focal loss over the detection head's raw NCHW outputs (the student head is
// re-run on the teacher features, so this loss runs twice per iteration over 8 x 201,600 anchors x 80 classes).
//   [ref: distillator.py:107-112 -> student.losses(...) -> detectron2 RetinaNet.losses -> fvcore
//    sigmoid_focal_loss_jit(alpha, gamma, reduction="sum"); same loss in thirdparty_heads/fcos.py:146-152]
// The library path permutes the (N, A*K, H, W) logits to (N, HWA, K) (a 516 MB copy, and its backward), builds a
// one-hot target of the same size and runs ~15 elementwise kernels over it.  Here the loss is evaluated in place
// on the NCHW tensor: element (n, a*K + k, y, x) is anchor (y, x, a) / class k, its target is
// [label(n, a, y, x) == k] from a small int32 label plane, ignored anchors (label < 0) contribute nothing.
//   forward : sum over valid anchors and classes (fp64 partials, fixed-order reduction)      -- reads logits once
//   backward: dlogits in the same NCHW layout, scaled by the upstream gradient (device scalar) -- read + write
//   fused (round 3): loss AND pre-scaled gradient in ONE pass -- the loss is divided by a normaliser that is known when it is
//   evaluated and its upstream gradient in the training step is 1, so dlogits = (1 / normaliser) dsum/dlogits can be written while the
//   logits stream for the sum (one exp / log / rcp per element serve both): read + write once instead of read, then read + write
//   (165 + 197 -> ~200 us per call at config 2).  The backward is lgd_scale_unless_one: nothing unless the upstream gradient != 1.
#include "common.h"

namespace lgd {

constexpr int kFocalChunk = 4096;

struct FocalArgs {
    const float* x[LGD_MAX_LEVELS];
    const int32_t* lab[LGD_MAX_LEVELS];   // (N, A, H, W) int32; K = background, < 0 = ignore
    float* gx[LGD_MAX_LEVELS];
    int HW[LGD_MAX_LEVELS], cpp[LGD_MAX_LEVELS], wave0[LGD_MAX_LEVELS + 1];
    int L, N, A, K, nwaves, nfin;
    float alpha, gamma;
    double* ws;        // [nwaves] partial sums | [nfin] block sums
    float* loss;
    const float* gscale;
    unsigned* bound;   // MODE 2, optional: receives the float bits of a bound of |grad| (below)
};

// One exp, one log and one reciprocal per element: e = exp(-|x|), p = sigmoid(x), softplus(+-x) share log1p(e).
// v_exp_f32 / v_log_f32 / v_rcp_f32 (~1 ulp) -- the loss is compared at 1e-5 relative.
struct FocalTerms { float p, sp_pos, sp_neg; };  // sigmoid(x), softplus(x), softplus(-x)
__device__ __forceinline__ FocalTerms focal_terms(float x) {
    const float e = __expf(-fabsf(x));
    const float inv = __frcp_rn(1.f + e);
    const float l1p = e < 1e-4f ? e * (1.f - 0.5f * e) : __logf(1.f + e);
    FocalTerms t;
    t.p = x >= 0.f ? inv : e * inv;
    t.sp_pos = fmaxf(x, 0.f) + l1p;
    t.sp_neg = fmaxf(-x, 0.f) + l1p;
    return t;
}
// fvcore: ce = BCEWithLogits(x, t); p_t = p*t + (1-p)*(1-t); loss = alpha_t * ce * (1 - p_t)^gamma
__device__ __forceinline__ float focal_elem(float x, bool t, float alpha, float gamma) {
    const FocalTerms f = focal_terms(x);
    const float ce = t ? f.sp_neg : f.sp_pos;
    const float q = t ? 1.f - f.p : f.p;  // 1 - p_t
    const float mod = gamma == 2.f ? q * q : powf(q, gamma);
    const float at = alpha >= 0.f ? (t ? alpha : 1.f - alpha) : 1.f;
    return at * ce * mod;
}
// d/dx of the above:  t=1: alpha (1-p)^g [g p log p - (1-p)] ; t=0: (1-alpha) p^g [p - g (1-p) log(1-p)]
__device__ __forceinline__ float focal_grad(float x, bool t, float alpha, float gamma) {
    const FocalTerms f = focal_terms(x);
    const float at = alpha >= 0.f ? (t ? alpha : 1.f - alpha) : 1.f;
    if (t) {
        const float q = 1.f - f.p, mod = gamma == 2.f ? q * q : powf(q, gamma);
        return at * mod * (gamma * f.p * (-f.sp_neg) - q);
    }
    const float mod = gamma == 2.f ? f.p * f.p : powf(f.p, gamma);
    return at * mod * (f.p + gamma * (1.f - f.p) * f.sp_pos);
}

// loss element and its derivative from one set of transcendentals
__device__ __forceinline__ void focal_both(float x, bool t, float alpha, float gamma, float& loss, float& grad) {
    const FocalTerms f = focal_terms(x);
    const float at = alpha >= 0.f ? (t ? alpha : 1.f - alpha) : 1.f;
    if (t) {
        const float q = 1.f - f.p, mod = gamma == 2.f ? q * q : powf(q, gamma);
        loss = at * f.sp_neg * mod;
        grad = at * mod * (gamma * f.p * (-f.sp_neg) - q);
    } else {
        const float mod = gamma == 2.f ? f.p * f.p : powf(f.p, gamma);
        loss = at * f.sp_pos * mod;
        grad = at * mod * (f.p + gamma * (1.f - f.p) * f.sp_pos);
    }
}

template <int MODE>  // 0 forward, 1 backward, 2 forward + gradient (pre-scaled by gscale[0])
__global__ __launch_bounds__(256) void focal_kernel(FocalArgs a) {
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (MODE == 2 && a.bound && blockIdx.x == 0 && threadIdx.x == 0) {
        // |d focal / dx| <= max(alpha, 1 - alpha) (1 + gamma / e):  t = 1: alpha (1-p)^g |g p log p - (1-p)|,  t = 0: (1-alpha) p^g |p - g (1-p) log(1-p)|,
        // and u |log u| <= 1 / e on (0, 1).  The magnitude tag of the gradient maps (the f16x2 scale of the class convolution's backward,
        // csrc/h2.hip) without a pass over them: within ~2^2 of the true maximum while positives exist, and exact bookkeeping is not needed --
        // the scale only has to keep the pair's 2^-22 window around the large elements.
        const float at = a.alpha >= 0.f ? fmaxf(a.alpha, 1.f - a.alpha) : 1.f;
        const float gsb = a.gscale ? fabsf(a.gscale[0]) : 1.f;
        *a.bound = __builtin_bit_cast(unsigned, at * (1.f + a.gamma * 0.36787945f) * gsb * 1.0001f);
    }
    if (w >= a.nwaves) return;
    const int lane = threadIdx.x & 63;
    int l = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) l += (i < a.L && w >= a.wave0[i]) ? 1 : 0;
    const int local = w - a.wave0[l];
    const int cpp = a.cpp[l], HW = a.HW[l];
    const int plane = local / cpp, chunk = local % cpp;   // plane = (n * A + an) * K + k
    const int k = plane % a.K, na = plane / a.K;
    const float* __restrict__ px = a.x[l] + (size_t)plane * HW;
    const int32_t* __restrict__ pl = a.lab[l] + (size_t)na * HW;
    float* __restrict__ pg = MODE != 0 ? a.gx[l] + (size_t)plane * HW : nullptr;
    const float gs = MODE != 0 ? (a.gscale ? a.gscale[0] : 1.f) : 0.f;
    const int e0 = chunk * kFocalChunk, e1 = min(HW, e0 + kFocalChunk);
    double acc = 0.0;
    if ((HW & 3) == 0) {
        for (int e = e0 + lane * 4; e < e1; e += 1024) {
            float4 v[4];
            int4 t[4];
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * 256;
                if (ee < e1) { v[u] = ldg_stream4(px + ee); t[u] = *reinterpret_cast<const int4*>(pl + ee); }
            }
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * 256;
                if (ee >= e1) continue;
                const float xs[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                const int ls[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
                float o[4], part = 0.f;
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == 0) part += ls[j] >= 0 ? focal_elem(xs[j], ls[j] == k, a.alpha, a.gamma) : 0.f;
                    else if (MODE == 1) o[j] = ls[j] >= 0 ? gs * focal_grad(xs[j], ls[j] == k, a.alpha, a.gamma) : 0.f;
                    else {
                        float lo = 0.f, gr = 0.f;
                        if (ls[j] >= 0) focal_both(xs[j], ls[j] == k, a.alpha, a.gamma, lo, gr);
                        part += lo; o[j] = gs * gr;
                    }
                }
                if (MODE != 1) acc += (double)part;
                if (MODE != 0) *reinterpret_cast<float4*>(pg + ee) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    } else {
        for (int e = e0 + lane; e < e1; e += 64 * 8) {
            float xs[8];
            int ls[8];
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ee = e + u * 64;
                if (ee < e1) { xs[u] = ldg_stream(px + ee); ls[u] = pl[ee]; }
            }
            float part = 0.f;
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ee = e + u * 64;
                if (ee >= e1) continue;
                if (MODE == 0) part += ls[u] >= 0 ? focal_elem(xs[u], ls[u] == k, a.alpha, a.gamma) : 0.f;
                else if (MODE == 1) pg[ee] = ls[u] >= 0 ? gs * focal_grad(xs[u], ls[u] == k, a.alpha, a.gamma) : 0.f;
                else {
                    float lo = 0.f, gr = 0.f;
                    if (ls[u] >= 0) focal_both(xs[u], ls[u] == k, a.alpha, a.gamma, lo, gr);
                    part += lo; pg[ee] = gs * gr;
                }
            }
            acc += (double)part;
        }
    }
    if (MODE != 1) {
        acc = wave_sum(acc);
        if (lane == 0) a.ws[w] = acc;
    }
}

// x *= g[0] unless g[0] == 1 (the usual case in a training step: every block leaves after one scalar load)
struct ScaleArgs { float* x[LGD_MAX_LEVELS]; long long n[LGD_MAX_LEVELS]; int L; };
__global__ __launch_bounds__(256) void scale_unless_one_kernel(ScaleArgs a, const float* g, unsigned* bound) {
    const float s = g[0];
    if (s == 1.f) return;
    if (bound && blockIdx.x == 0 && threadIdx.x == 0)   // the maps' magnitude bound moves with them
        *bound = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, *bound) * fabsf(s) * 1.0001f);
    for (int l = 0; l < a.L; ++l) {
        float* x = a.x[l];
        const long long n = a.n[l], n4 = (reinterpret_cast<size_t>(x) & 15) == 0 ? n >> 2 : 0;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            float4 v = reinterpret_cast<float4*>(x)[i];
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
            reinterpret_cast<float4*>(x)[i] = v;
        }
        for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] *= s;
    }
}

__global__ __launch_bounds__(256) void focal_reduce_kernel(FocalArgs a, int stage) {
    __shared__ double red[4];
    // stage 0: nwaves partials -> nfin block sums ; stage 1: nfin block sums -> loss
    const double* src = stage == 0 ? a.ws : a.ws + a.nwaves;
    const int n = stage == 0 ? a.nwaves : a.nfin;
    double s = 0.0;
    if (stage == 0) {
        const int i0 = blockIdx.x * 4096;
        for (int i = i0 + threadIdx.x; i < min(n, i0 + 4096); i += 256) s += src[i];
    } else {
        for (int i = threadIdx.x; i < n; i += 256) s += src[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (red[0] + red[1]) + (red[2] + red[3]);
        if (stage == 0) a.ws[a.nwaves + blockIdx.x] = tot;
        else a.loss[0] = (float)tot;
    }
}

static int focal_fill(FocalArgs& a, const float* const* x_host, const int32_t* const* lab_host, const int32_t* level_hw_host,
                      int L, int N, int A, int K, float alpha, float gamma) {
    if (!x_host || !lab_host || !level_hw_host || L < 1 || L > LGD_MAX_LEVELS || N < 1 || A < 1 || K < 1) return LGD_EINVAL;
    a.L = L; a.N = N; a.A = A; a.K = K; a.alpha = alpha; a.gamma = gamma;
    int w = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.x[l] = nullptr; a.lab[l] = nullptr; a.gx[l] = nullptr;
        a.wave0[l] = w;
        if (l < L) {
            if (!x_host[l] || !lab_host[l]) return LGD_EINVAL;
            a.x[l] = x_host[l]; a.lab[l] = lab_host[l];
            a.HW[l] = level_hw_host[2 * l] * level_hw_host[2 * l + 1];
            a.cpp[l] = (a.HW[l] + kFocalChunk - 1) / kFocalChunk;
            w += N * A * K * a.cpp[l];
        } else { a.HW[l] = 0; a.cpp[l] = 1; }
    }
    a.wave0[LGD_MAX_LEVELS] = w;
    a.nwaves = w;
    a.nfin = (w + 4095) / 4096;
    return LGD_OK;
}

}  // namespace lgd

extern "C" {

size_t lgd_focal_ws_doubles(const int32_t* level_hw_host, int L, int N, int A, int K) {
    size_t w = 0;
    for (int l = 0; l < L; ++l) {
        const int hw = level_hw_host[2 * l] * level_hw_host[2 * l + 1];
        w += (size_t)N * A * K * ((hw + lgd::kFocalChunk - 1) / lgd::kFocalChunk);
    }
    return w + (w + 4095) / 4096;
}

int lgd_focal_loss_fwd(const float* const* logits_host, const int32_t* const* labels_host, const int32_t* level_hw_host, int L,
                       int N, int A, int K, float alpha, float gamma, double* ws, float* loss, void* stream) {
    lgd::FocalArgs a;
    if (lgd::focal_fill(a, logits_host, labels_host, level_hw_host, L, N, A, K, alpha, gamma) != LGD_OK || !ws || !loss)
        return LGD_EINVAL;
    a.ws = ws; a.loss = loss; a.gscale = nullptr; a.bound = nullptr;
    hipStream_t s = (hipStream_t)stream;
    LGD_LAUNCH("focal_fwd_kernel", lgd::focal_kernel<0>, dim3((a.nwaves + 3) / 4), dim3(256), 0, s, a);
    LGD_LAUNCH("focal_reduce_kernel", lgd::focal_reduce_kernel, dim3(a.nfin), dim3(256), 0, s, a, 0);
    LGD_LAUNCH("focal_reduce_kernel", lgd::focal_reduce_kernel, dim3(1), dim3(256), 0, s, a, 1);
    return lgd::check_launch();
}

int lgd_focal_loss_fwd_grad(const float* const* logits_host, const int32_t* const* labels_host, const int32_t* level_hw_host, int L,
                            int N, int A, int K, float alpha, float gamma, const float* grad_scale, double* ws, float* loss,
                            float* const* grad_logits_host, uint32_t* bound_out, void* stream) {
    lgd::FocalArgs a;
    if (lgd::focal_fill(a, logits_host, labels_host, level_hw_host, L, N, A, K, alpha, gamma) != LGD_OK || !ws || !loss || !grad_logits_host)
        return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!grad_logits_host[l]) return LGD_EINVAL; a.gx[l] = grad_logits_host[l]; }
    a.ws = ws; a.loss = loss; a.gscale = grad_scale; a.bound = bound_out;
    hipStream_t s = (hipStream_t)stream;
    LGD_LAUNCH("focal_fwd_grad_kernel", lgd::focal_kernel<2>, dim3((a.nwaves + 3) / 4), dim3(256), 0, s, a);
    LGD_LAUNCH("focal_reduce_kernel", lgd::focal_reduce_kernel, dim3(a.nfin), dim3(256), 0, s, a, 0);
    LGD_LAUNCH("focal_reduce_kernel", lgd::focal_reduce_kernel, dim3(1), dim3(256), 0, s, a, 1);
    return lgd::check_launch();
}

int lgd_scale_unless_one(float* const* x_host, const long long* n_host, int L, const float* g, uint32_t* bound_inout, void* stream) {
    if (!x_host || !n_host || !g || L < 1 || L > LGD_MAX_LEVELS) return LGD_EINVAL;
    lgd::ScaleArgs a;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) { a.x[l] = nullptr; a.n[l] = 0; }
    for (int l = 0; l < L; ++l) { if (!x_host[l] || n_host[l] < 0) return LGD_EINVAL; a.x[l] = x_host[l]; a.n[l] = n_host[l]; }
    a.L = L;
    LGD_LAUNCH("scale_unless_one_kernel", lgd::scale_unless_one_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, a, g, bound_inout);
    return lgd::check_launch();
}

int lgd_focal_loss_bwd(const float* const* logits_host, const int32_t* const* labels_host, const int32_t* level_hw_host, int L,
                       int N, int A, int K, float alpha, float gamma, const float* grad_loss, float* const* grad_logits_host,
                       void* stream) {
    lgd::FocalArgs a;
    if (lgd::focal_fill(a, logits_host, labels_host, level_hw_host, L, N, A, K, alpha, gamma) != LGD_OK || !grad_loss ||
        !grad_logits_host)
        return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!grad_logits_host[l]) return LGD_EINVAL; a.gx[l] = grad_logits_host[l]; }
    a.ws = nullptr; a.loss = nullptr; a.gscale = grad_loss; a.bound = nullptr;
    LGD_LAUNCH("focal_bwd_kernel", lgd::focal_kernel<1>, dim3((a.nwaves + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
