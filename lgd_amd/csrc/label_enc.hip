// K6: label-encoder glue -- box descriptors, row LayerNorm(+ReLU), per-row vector x matrix (T-Net transform),
// per-image max pooling.  With lgd_gemm_batch for the pointwise convolutions / FC layers this covers the whole
// LabelEncoder + STN  [ref: dynamic_teacher/label_encoder.py:12-115 (descriptors), 216-276 (PointNet),
// dynamic_teacher/spatial_transformer.py:30-47].  T (boxes in the mini-batch) is ~10^2: every kernel here is
// latency-bound; the point is fewer, fatter launches with no host round trips (the reference does a .tolist()
// per image and an .item() per pooling), not bandwidth.
#include "common.h"

#pragma clang fp contract(off)  // descriptors must round like the reference's separate torch ops

namespace lgd {

// ------------------------------------------------------------------------------------------- descriptors
struct DescArgs {
    const float* boxes_in;     // (T0,4) instance boxes as given (x1y1x2y2 or x1y1wh), image-major
    const int32_t* classes;    // (T0,)
    const int32_t* in_off;     // (B+1) instance offsets
    const int32_t* out_off;    // (B+1) row offsets of the output (instances [+ctx] or 1 substitute row)
    float* desc;               // (T, 4+K) in [-1,1]
    float* boxes_out;          // (T,4) clamped xyxy in padded-image pixels (the reference's `boxlists`)
    int B, T, K, img_h, img_w, add_ctx, wh_format;
};

__global__ __launch_bounds__(256) void box_desc_kernel(DescArgs a) {
    const int D = 4 + a.K;
    // one wave per output row
    const int t = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (t >= a.T) return;
    const int lane = threadIdx.x & 63;
    int b = 0;
    while (b + 1 < a.B && t >= a.out_off[b + 1]) ++b;
    const int j = t - a.out_off[b];
    const int i0 = a.in_off[b], n = a.in_off[b + 1] - i0;
    float x1, y1, x2, y2;
    int cls = -1;
    if (n == 0) {                       // label_encoder.py:64-66 substitute box, class vector all zero
        x1 = 0.f; y1 = 0.f; x2 = 1.f; y2 = 1.f;
        if (a.wh_format) { x2 = (x1 + 1.f) - 1.f; y2 = (y1 + 1.f) - 1.f; }
    } else if (j == n) {                // label_encoder.py:75-77 context box, class vector all zero
        x1 = 0.f; y1 = 0.f; x2 = (float)a.img_w; y2 = (float)a.img_h;
    } else {
        const float4 v = reinterpret_cast<const float4*>(a.boxes_in)[i0 + j];
        x1 = v.x; y1 = v.y; x2 = v.z; y2 = v.w;
        if (a.wh_format) { x2 = (x1 + v.z) - 1.f; y2 = (y1 + v.w) - 1.f; }  // utils.py:26-38
        cls = a.classes[i0 + j];
    }
    const float mw = (float)(a.img_w - 1), mh = (float)(a.img_h - 1);   // utils.py:40-51
    x1 = fminf(fmaxf(x1, 0.f), mw); x2 = fminf(fmaxf(x2, 0.f), mw);
    y1 = fminf(fmaxf(y1, 0.f), mh); y2 = fminf(fmaxf(y2, 0.f), mh);
    if (lane == 0) reinterpret_cast<float4*>(a.boxes_out)[t] = make_float4(x1, y1, x2, y2);
    const float nb[4] = {x1 / (float)a.img_w, y1 / (float)a.img_h, x2 / (float)a.img_w, y2 / (float)a.img_h};   // label_encoder.py:88-89
    for (int d = lane; d < D; d += 64) {
        const float v = d < 4 ? nb[d] : ((d - 4) == cls ? 1.f : 0.f);
        a.desc[(size_t)t * D + d] = 2.f * v - 1.f;  // plain operators under contract(off): one rounding per torch op  // range_scaling [0,1] -> [-1,1]
    }
}

// ------------------------------------------------------------------------------------------- row LayerNorm (+ReLU)
// y = relu?((x - mean) * rstd) over the feature axis (no affine, eps 1e-5, biased variance); one wave per row.
// stats (T,2) = mean, rstd kept for the backward; backward: dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),
// g = dy * [y > 0] when relu.
template <int MODE>
__global__ __launch_bounds__(256) void rowln_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out,
                                                    float* __restrict__ stats, int T, int F, int relu) {
    const int t = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (t >= T) return;
    const int lane = threadIdx.x & 63;
    const float* px = x + (size_t)t * F;
    if (MODE == 0) {
        float s = 0.f;
        for (int f = lane; f < F; f += 64) s += px[f];
        const float mean = wave_sum(s) / (float)F;
        float q = 0.f;
        for (int f = lane; f < F; f += 64) { const float d = px[f] - mean; q = fmaf(d, d, q); }
        const float rstd = rsqrtf(wave_sum(q) / (float)F + 1e-5f);
        if (lane == 0) { stats[2 * t] = mean; stats[2 * t + 1] = rstd; }
        for (int f = lane; f < F; f += 64) {
            const float v = (px[f] - mean) * rstd;
            out[(size_t)t * F + f] = relu ? fmaxf(v, 0.f) : v;
        }
    } else {
        const float mean = stats[2 * t], rstd = stats[2 * t + 1];
        const float* pd = dy + (size_t)t * F;
        float s1 = 0.f, s2 = 0.f;
        for (int f = lane; f < F; f += 64) {
            const float xh = (px[f] - mean) * rstd;
            const float g = (relu && !(xh > 0.f)) ? 0.f : pd[f];
            s1 += g; s2 = fmaf(g, xh, s2);
        }
        const float m1 = wave_sum(s1) / (float)F, m2 = wave_sum(s2) / (float)F;
        for (int f = lane; f < F; f += 64) {
            const float xh = (px[f] - mean) * rstd;
            const float g = (relu && !(xh > 0.f)) ? 0.f : pd[f];
            out[(size_t)t * F + f] = rstd * (g - m1 - xh * m2);
        }
    }
}

// ------------------------------------------------------------------------------------------- row vector x matrix
// out[t, j] = sum_i x[t, i] * M[t, i, j]   (the T-Net transform, label_encoder.py:241,248: bmm(x^T, M)^T)
// one workgroup per row; thread j owns output column j (M rows are read coalesced).
__global__ __launch_bounds__(128) void rowvecmat_fwd_kernel(const float* __restrict__ x, const float* __restrict__ M, float* __restrict__ out, int k) {
    const int t = blockIdx.x, j = threadIdx.x;
    __shared__ float sx[128];
    if (j < k) sx[j] = x[(size_t)t * k + j];
    __syncthreads();
    if (j >= k) return;
    const float* m = M + (size_t)t * k * k;
    float acc = 0.f;
    #pragma unroll 4
    for (int i = 0; i < k; ++i) acc = fmaf(sx[i], m[(size_t)i * k + j], acc);
    out[(size_t)t * k + j] = acc;
}
// dx[t, i] = sum_j dout[t, j] M[t, i, j] ; dM[t, i, j] = x[t, i] dout[t, j]
__global__ __launch_bounds__(128) void rowvecmat_bwd_kernel(const float* __restrict__ x, const float* __restrict__ M, const float* __restrict__ dout,
                                                            float* __restrict__ dx, float* __restrict__ dM, int k) {
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float sx[128], sd[128];
    if (tid < k) { sx[tid] = x[(size_t)t * k + tid]; sd[tid] = dout[(size_t)t * k + tid]; }
    __syncthreads();
    const float* m = M + (size_t)t * k * k;
    float* gm = dM + (size_t)t * k * k;
    for (int i = wave; i < k; i += 2) {  // each wave owns rows i of M: coalesced over j
        float part = 0.f;
        const float xi = sx[i];
        for (int j = lane; j < k; j += 64) {
            part = fmaf(sd[j], m[(size_t)i * k + j], part);
            gm[(size_t)i * k + j] = xi * sd[j];
        }
        part = wave_sum(part);
        if (lane == 0) dx[(size_t)t * k + i] = part;
    }
}

// ------------------------------------------------------------------------------------------- per-image max pooling
// out[t, f] = max over the rows of image(t) of x[., f]  (label_encoder.py:195-213 hier_pool + 262-264 repeat);
// arg (B,F) = row index of the maximum (first occurrence), kept for the backward:
// dx[t, f] = sum over the rows of image(t) of dout[., f] if t is the argmax row, else 0.
__global__ __launch_bounds__(256) void segmax_fwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ off, float* __restrict__ out,
                                                         int32_t* __restrict__ arg, int F) {
    const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const int t0 = off[b], t1 = off[b + 1];
    float best = -INFINITY;
    int bi = t0;
    for (int t = t0; t < t1; ++t) {
        const float v = x[(size_t)t * F + f];
        if (v > best) { best = v; bi = t; }
    }
    arg[(size_t)b * F + f] = bi;
    for (int t = t0; t < t1; ++t) out[(size_t)t * F + f] = best;
}
__global__ __launch_bounds__(256) void segmax_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ off, const int32_t* __restrict__ arg,
                                                         float* __restrict__ dx, int F) {
    const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const int t0 = off[b], t1 = off[b + 1];
    float s = 0.f;
    for (int t = t0; t < t1; ++t) s += dout[(size_t)t * F + f];
    const int bi = arg[(size_t)b * F + f];
    for (int t = t0; t < t1; ++t) dx[(size_t)t * F + f] = t == bi ? s : 0.f;
}

}  // namespace lgd

extern "C" {

int lgd_box_descriptors(const float* boxes_in, const int32_t* classes, const int32_t* in_off, const int32_t* out_off, int B, int T,
                        int num_classes, int img_h, int img_w, int add_ctx, int wh_format, float* desc, float* boxes_out,
                        void* stream) {
    if (!in_off || !out_off || !desc || !boxes_out || B < 1 || T < 1 || num_classes < 1) return LGD_EINVAL;
    lgd::DescArgs a;
    a.boxes_in = boxes_in; a.classes = classes; a.in_off = in_off; a.out_off = out_off; a.desc = desc; a.boxes_out = boxes_out;
    a.B = B; a.T = T; a.K = num_classes; a.img_h = img_h; a.img_w = img_w; a.add_ctx = add_ctx; a.wh_format = wh_format;
    LGD_LAUNCH("box_desc_kernel", lgd::box_desc_kernel, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

int lgd_rowln_fwd(const float* x, int T, int F, int relu, float* y, float* stats, void* stream) {
    if (!x || !y || !stats || T < 1 || F < 1) return LGD_EINVAL;
    LGD_LAUNCH("rowln_kernel", lgd::rowln_kernel<0>, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, y, stats, T, F, relu);
    return lgd::check_launch();
}
int lgd_rowln_bwd(const float* x, const float* dy, const float* stats, int T, int F, int relu, float* dx, void* stream) {
    if (!x || !dy || !dx || !stats || T < 1 || F < 1) return LGD_EINVAL;
    LGD_LAUNCH("rowln_bwd_kernel", lgd::rowln_kernel<1>, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, dx, const_cast<float*>(stats), T, F, relu);
    return lgd::check_launch();
}

int lgd_rowvecmat_fwd(const float* x, const float* M, int T, int k, float* out, void* stream) {
    if (!x || !M || !out || T < 1 || k < 1 || k > 128) return LGD_EINVAL;
    LGD_LAUNCH("rowvecmat_fwd_kernel", lgd::rowvecmat_fwd_kernel, dim3(T), dim3(128), 0, (hipStream_t)stream, x, M, out, k);
    return lgd::check_launch();
}
int lgd_rowvecmat_bwd(const float* x, const float* M, const float* dout, int T, int k, float* dx, float* dM, void* stream) {
    if (!x || !M || !dout || !dx || !dM || T < 1 || k < 1 || k > 128) return LGD_EINVAL;
    LGD_LAUNCH("rowvecmat_bwd_kernel", lgd::rowvecmat_bwd_kernel, dim3(T), dim3(128), 0, (hipStream_t)stream, x, M, dout, dx, dM, k);
    return lgd::check_launch();
}

int lgd_segmax_fwd(const float* x, const int32_t* off, int B, int F, float* out, int32_t* arg, void* stream) {
    if (!x || !off || !out || !arg || B < 1 || F < 1) return LGD_EINVAL;
    LGD_LAUNCH("segmax_fwd_kernel", lgd::segmax_fwd_kernel, dim3((F + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, off, out, arg, F);
    return lgd::check_launch();
}
int lgd_segmax_bwd(const float* dout, const int32_t* off, const int32_t* arg, int B, int F, float* dx, void* stream) {
    if (!dout || !off || !arg || !dx || B < 1 || F < 1) return LGD_EINVAL;
    LGD_LAUNCH("segmax_bwd_kernel", lgd::segmax_bwd_kernel, dim3((F + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dout, off, arg, dx, F);
    return lgd::check_launch();
}

}  // extern "C"
