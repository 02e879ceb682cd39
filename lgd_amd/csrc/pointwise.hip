// Epilogues of the student's library convolutions [ref: the detectron2 BottleneckBlock the reference builds its student
// from: out = conv3(...) (FrozenBN folded to a per-channel bias); out += shortcut; relu].
// torch runs bias add, residual add and ReLU as three passes (7 map transfers); here one kernel reads the conv output
// [+ the residual] and writes relu(x + bias[c] + r) once (2-3 transfers), and one kernel applies the ReLU mask of the
// saved output to the incoming gradient.  Pure HBM streaming: thread per float4, plane index from the flat offset.
#include "common.h"

namespace lgd {

struct BiasActArgs {
    const float* x;      // (N, C, HW) conv output (no bias)
    const float* bias;   // [C] or null
    const float* res;    // (N, C, HW) residual or null
    float* out;
    uint32_t* bits;      // null, or the linear bitmap [out > 0]: element e <-> bit e % 32 of word e / 32 (the backward's ReLU mask)
    long long total;     // N*C*HW
    int C, HW, relu;
};

template <int VW>
__global__ __launch_bounds__(256) void bias_act_kernel(BiasActArgs a) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * VW;
    const bool on = i < a.total;
    Vec<VW> v;
    #pragma unroll
    for (int k = 0; k < VW; ++k) v.v[k] = 0.f;
    if (on) {
        const int c = (int)((i / a.HW) % a.C);  // VW divides HW: the VW elements share a plane
        const float b = a.bias ? a.bias[c] : 0.f;
        v = vload<VW>(a.x + i);
        if (a.res) {
            const Vec<VW> r = vload<VW>(a.res + i);
            #pragma unroll
            for (int k = 0; k < VW; ++k) v.v[k] += r.v[k];
        }
        #pragma unroll
        for (int k = 0; k < VW; ++k) {
            v.v[k] += b;
            if (a.relu) v.v[k] = fmaxf(v.v[k], 0.f);
        }
        vstore<VW>(a.out + i, v);
    }
    if (a.bits) {   // wave-uniform: the backward reads 1 bit instead of the 4-byte output
        if constexpr (VW == 4) {   // a nibble per lane, eight lanes to a word (i of lane 8j is a multiple of 32)
            unsigned w = 0u;
            #pragma unroll
            for (int k = 0; k < 4; ++k) w |= (v.v[k] > 0.f ? 1u : 0u) << k;
            w <<= (threadIdx.x & 7) * 4;
            w |= __shfl_xor(w, 1); w |= __shfl_xor(w, 2); w |= __shfl_xor(w, 4);
            if ((threadIdx.x & 7) == 0 && on) a.bits[i >> 5] = w;
        } else {                   // a bit per lane: the wave's ballot is two words (i of lane 0 is a multiple of 64)
            const unsigned long long m = __ballot(v.v[0] > 0.f);
            if ((threadIdx.x & 63) == 0 && on) { a.bits[i >> 5] = (uint32_t)m; a.bits[(i >> 5) + 1] = (uint32_t)(m >> 32); }
        }
    }
}

// dx = dy where the forward output was > 0, from the bitmap bias_act wrote: 2 map transfers instead of 3
template <int VW>
__global__ __launch_bounds__(256) void relu_bits_kernel(const uint32_t* __restrict__ bits, const float* __restrict__ dy,
                                                        float* __restrict__ dx, long long total) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * VW;
    if (i >= total) return;
    const unsigned w = bits[i >> 5] >> (unsigned)(i & 31);
    const Vec<VW> g = vload<VW>(dy + i);
    Vec<VW> r;
    #pragma unroll
    for (int k = 0; k < VW; ++k) r.v[k] = ((w >> k) & 1u) ? g.v[k] : 0.f;
    vstore<VW>(dx + i, r);
}

// the same from the row-padded bitmap the gemm3 epilogue writes (csrc/gemm3.hip: a word per 32 columns of a row, rows = (image, channel)
// planes of HW elements): thread per VW consecutive columns of one row (VW | 32: one word)
template <int VW>
__global__ __launch_bounds__(256) void relu_rowbits_kernel(const uint32_t* __restrict__ bits, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int HW, int wpr) {
    const int n = (blockIdx.y * 256 + threadIdx.x) * VW;
    if (n >= HW) return;
    const long long row = blockIdx.x;
    const unsigned w = bits[row * wpr + (n >> 5)] >> (unsigned)(n & 31);
    const long long i = row * HW + n;
    const Vec<VW> g = vload<VW>(dy + i);
    Vec<VW> r;
    #pragma unroll
    for (int k = 0; k < VW; ++k) r.v[k] = ((w >> k) & 1u) ? g.v[k] : 0.f;
    vstore<VW>(dx + i, r);
}

// ... leaving max |dx| as well (float bits, atomic max onto a word the caller zeroed): the gradient that reaches a block's last ReLU is often a sum
// autograd built, which carries no magnitude tag -- with the bound from here the 1x1 input- and weight-gradient products that read dx take their
// f16x2 forms (csrc/gemm3.hip lgd_gemm2h, csrc/h2.hip lgd_h2_pwdw).  Four chunks per thread: a quarter of the workgroups, one report each.
template <int VW>
__global__ __launch_bounds__(256) void relu_rowbits_amax_kernel(const uint32_t* __restrict__ bits, const float* __restrict__ dy,
                                                                float* __restrict__ dx, int HW, int wpr, unsigned* __restrict__ amax) {
    __shared__ float slots[4];
    const long long row = blockIdx.x;
    float am = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int n = ((blockIdx.y * 4 + it) * 256 + threadIdx.x) * VW;
        if (n < HW) {
            const unsigned w = bits[row * wpr + (n >> 5)] >> (unsigned)(n & 31);
            const long long i = row * HW + n;
            const Vec<VW> g = vload<VW>(dy + i);
            Vec<VW> r;
#pragma unroll
            for (int k = 0; k < VW; ++k) { r.v[k] = ((w >> k) & 1u) ? g.v[k] : 0.f; am = fmaxf(am, fabsf(r.v[k])); }
            vstore<VW>(dx + i, r);
        }
    }
    block_max_bits(amax, wave_max(am), slots);
}

template <int VW>
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                        float* __restrict__ dx, long long total) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * VW;
    if (i >= total) return;
    const Vec<VW> o = vload<VW>(y + i), g = vload<VW>(dy + i);
    Vec<VW> r;
    #pragma unroll
    for (int k = 0; k < VW; ++k) r.v[k] = o.v[k] > 0.f ? g.v[k] : 0.f;
    vstore<VW>(dx + i, r);
}

// every other pixel of every other row (the input of a 1x1 / stride 2 convolution) and its adjoint: one pass each.  torch runs
// x[:, :, ::2, ::2].contiguous() as a strided copy and its backward as two zero fills + two strided copies over the full-resolution map
// (res3 -> res4 at BASELINE config 2: 170 us per step for the backward alone)
__global__ __launch_bounds__(256) void subsample2_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H, int W, int Ho, int Wo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= planes * Ho * Wo) return;
    const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
    const long long p = i / ((long long)Wo * Ho);
    y[i] = x[(p * H + 2 * yo) * W + 2 * xo];
}
// thread per pair of input columns (x = 2 xo, 2 xo + 1) of one row: (g, 0) on even rows, (0, 0) on odd ones
__global__ __launch_bounds__(256) void subsample2_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, long long planes, int H, int W, int Ho, int Wo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= planes * H * Wo) return;
    const int xo = (int)(i % Wo), yy = (int)((i / Wo) % H);
    const long long p = i / ((long long)Wo * H);
    const float v = (yy & 1) ? 0.f : g[(p * Ho + (yy >> 1)) * Wo + xo];
    float* o = dx + (p * H + yy) * W + 2 * xo;
    if (2 * xo + 1 < W) {
        if (((uintptr_t)o & 7) == 0) *reinterpret_cast<float2*>(o) = make_float2(v, 0.f);
        else { o[0] = v; o[1] = 0.f; }
    } else {
        o[0] = v;
    }
}

}  // namespace lgd

extern "C" {

int lgd_subsample2_fwd(const float* x, long long planes, int H, int W, float* y, void* stream) {
    if (!x || !y || planes < 1 || H < 1 || W < 1) return LGD_EINVAL;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = planes * Ho * Wo;
    LGD_LAUNCH("subsample2_kernel", lgd::subsample2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, planes, H, W, Ho, Wo);
    return lgd::check_launch();
}

int lgd_subsample2_bwd(const float* g, long long planes, int H, int W, float* dx, void* stream) {
    if (!g || !dx || planes < 1 || H < 1 || W < 1) return LGD_EINVAL;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = planes * H * Wo;
    LGD_LAUNCH("subsample2_bwd_kernel", lgd::subsample2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, dx, planes, H, W, Ho, Wo);
    return lgd::check_launch();
}

size_t lgd_relu_bits_words(long long total) { return total < 1 ? 0 : (size_t)((total + 63) / 64) * 2; }

int lgd_bias_act_fwd(const float* x, const float* bias, const float* residual, int N, int C, int HW, int relu, float* out,
                     uint32_t* relu_bits, void* stream) {
    if (!x || !out || N < 1 || C < 1 || HW < 1) return LGD_EINVAL;
    lgd::BiasActArgs a{x, bias, residual, out, relu_bits, (long long)N * C * HW, C, HW, relu ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    const bool al = (((uintptr_t)x | (uintptr_t)out | (uintptr_t)(residual ? residual : x)) & 15) == 0;
    if (HW % 4 == 0 && al) {
        LGD_LAUNCH("bias_act_kernel", lgd::bias_act_kernel<4>, dim3((unsigned)((a.total / 4 + 255) / 256)), dim3(256), 0, st, a);
    } else {
        LGD_LAUNCH("bias_act_kernel", lgd::bias_act_kernel<1>, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
    }
    return lgd::check_launch();
}

int lgd_relu_bits_bwd(const uint32_t* relu_bits, const float* dy, long long total, float* dx, void* stream) {
    if (!relu_bits || !dy || !dx || total < 1) return LGD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool al = (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
    if (total % 4 == 0 && al) {
        LGD_LAUNCH("relu_bits_kernel", lgd::relu_bits_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, relu_bits, dy, dx, total);
    } else {
        LGD_LAUNCH("relu_bits_kernel", lgd::relu_bits_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, relu_bits, dy, dx, total);
    }
    return lgd::check_launch();
}

size_t lgd_relu_rowbits_words(long long rows, int HW) { return rows < 1 || HW < 1 ? 0 : (size_t)rows * (size_t)((HW + 31) / 32); }

int lgd_relu_rowbits_bwd(const uint32_t* relu_bits, const float* dy, long long rows, int HW, float* dx, uint32_t* amax_out, void* stream) {
    if (!relu_bits || !dy || !dx || rows < 1 || rows >= (1LL << 31) || HW < 1) return LGD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool al = (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
    const int wpr = (HW + 31) / 32;
    if (amax_out) {
        if (HW % 4 == 0 && al) {
            LGD_LAUNCH("relu_rowbits_kernel", lgd::relu_rowbits_amax_kernel<4>, dim3((unsigned)rows, (unsigned)((HW / 4 + 1023) / 1024)), dim3(256), 0, st, relu_bits, dy, dx, HW, wpr, amax_out);
        } else {
            LGD_LAUNCH("relu_rowbits_kernel", lgd::relu_rowbits_amax_kernel<1>, dim3((unsigned)rows, (unsigned)((HW + 1023) / 1024)), dim3(256), 0, st, relu_bits, dy, dx, HW, wpr, amax_out);
        }
        return lgd::check_launch();
    }
    if (HW % 4 == 0 && al) {
        LGD_LAUNCH("relu_rowbits_kernel", lgd::relu_rowbits_kernel<4>, dim3((unsigned)rows, (unsigned)((HW / 4 + 255) / 256)), dim3(256), 0, st, relu_bits, dy, dx, HW, wpr);
    } else {
        LGD_LAUNCH("relu_rowbits_kernel", lgd::relu_rowbits_kernel<1>, dim3((unsigned)rows, (unsigned)((HW + 255) / 256)), dim3(256), 0, st, relu_bits, dy, dx, HW, wpr);
    }
    return lgd::check_launch();
}

int lgd_relu_mask_bwd(const float* y, const float* dy, long long total, float* dx, void* stream) {
    if (!y || !dy || !dx || total < 1) return LGD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool al = (((uintptr_t)y | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
    if (total % 4 == 0 && al) {
        LGD_LAUNCH("relu_mask_kernel", lgd::relu_mask_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, y, dy, dx, total);
    } else {
        LGD_LAUNCH("relu_mask_kernel", lgd::relu_mask_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, dy, dx, total);
    }
    return lgd::check_launch();
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// Frozen stem epilogue: out = max_pool2d(relu(y + bias[c]), 3, stride 2, padding 1) of the 7x7 / stride-2 stem convolution's output y
// [d2-memory: BasicStem.forward -- conv1 -> FrozenBN -> relu_ -> max_pool2d(kernel 3, stride 2, padding 1); SURVEY.md appendix A].
// relu and the bias add are monotonic, so max(relu(v + b)) = relu(max(v) + b): one pass reads the conv output once (through the
// caches for the overlapping windows) and writes the quarter-size map.  torch: bias / ReLU pass (2 maps) + pooling kernel (1.25
// maps at 1.7 TB/s): 0.2 + 0.4 ms per step at config 2 for a map that is 550 MB.  Forward only: the stem is frozen (FREEZE_AT >= 1).
namespace lgd {

__global__ __launch_bounds__(256) void stem_pool_kernel(const float* __restrict__ y, const float* __restrict__ bias, float* __restrict__ out,
                                                        int C, int H, int W, int Ho, int Wo, long long total) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int ox = (int)(o % Wo), oy = (int)((o / Wo) % Ho);
    const long long plane = o / ((long long)Wo * Ho);            // n * C + c
    const float* p = y + plane * H * W;
    const int x0 = 2 * ox - 1, y0 = 2 * oy - 1;
    float m = -INFINITY;                                          // the padding never wins: the window always holds (2 oy, 2 ox)
    #pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int yy = y0 + i;
        if (yy < 0 || yy >= H) continue;
        const float* row = p + (size_t)yy * W;
        #pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int xx = x0 + j;
            if (xx >= 0 && xx < W) m = fmaxf(m, row[xx]);   // plain loads: neighbouring windows re-read these lines from the caches
        }
    }
    out[o] = fmaxf(m + bias[(int)(plane % C)], 0.f);
}

// W % 4 == 0: a thread produces TWO horizontally adjacent outputs from one aligned float4 per input row (columns 4k .. 4k+3) and the
// column 4k-1 from its left neighbour lane (a DPP shift; the wave's first lane loads it): fully coalesced 16-byte loads instead of
// nine stride-2 dword loads per output (measured 319 -> 170 us on the 8 x 64 x 400 x 672 stem map; torch: ~600 us for the two passes).
__global__ __launch_bounds__(256) void stem_pool_pair_kernel(const float* __restrict__ y, const float* __restrict__ bias, float* __restrict__ out,
                                                             int C, int H, int W, int Ho, int Wo, long long total_pairs) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool on = u < total_pairs;
    const long long uu = on ? u : total_pairs - 1;
    const int WP = Wo >> 1;
    const int k = (int)(uu % WP), oy = (int)((uu / WP) % Ho);
    const long long plane = uu / ((long long)WP * Ho);
    const float* p = y + plane * H * W;
    const int lane = threadIdx.x & 63;
    float m0 = -INFINITY, m1 = -INFINITY;
    #pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int yy = 2 * oy - 1 + i;
        const bool yok = yy >= 0 && yy < H;
        const float* row = p + (size_t)(yok ? yy : 0) * W + 4 * k;
        float4 v = *reinterpret_cast<const float4*>(row);
        float left = wave_shr1(v.w);                       // column 4k-1 = the previous pair's last column (same row unless k == 0)
        if (lane == 0 && k > 0) left = row[-1];
        if (k == 0) left = -INFINITY;
        if (yok) {
            m0 = fmaxf(m0, fmaxf(fmaxf(left, v.x), v.y));   // output 2k  : columns 4k-1, 4k, 4k+1
            m1 = fmaxf(m1, fmaxf(fmaxf(v.y, v.z), v.w));    // output 2k+1: columns 4k+1, 4k+2, 4k+3
        }
    }
    if (!on) return;
    const float b = bias[(int)(plane % C)];
    float2 r = make_float2(fmaxf(m0 + b, 0.f), fmaxf(m1 + b, 0.f));
    *reinterpret_cast<float2*>(out + (plane * Ho + oy) * (long long)Wo + 2 * k) = r;
}

}  // namespace lgd

extern "C" int lgd_stem_bias_relu_maxpool(const float* y, const float* bias, int N, int C, int H, int W, float* out, void* stream) {
    if (!y || !bias || !out || N < 1 || C < 1 || H < 1 || W < 1) return LGD_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;         // floor((H + 2 - 3) / 2) + 1
    const long long total = (long long)N * C * Ho * Wo;
    if (W % 4 == 0 && (((uintptr_t)y | (uintptr_t)out) & 15) == 0) {   // Wo = W / 2 is even, rows of y / out stay 16- / 8-byte aligned
        const long long pairs = total / 2;
        LGD_LAUNCH("stem_pool_kernel", lgd::stem_pool_pair_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0,
                   (hipStream_t)stream, y, bias, out, C, H, W, Ho, Wo, pairs);
        return lgd::check_launch();
    }
    LGD_LAUNCH("stem_pool_kernel", lgd::stem_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
               y, bias, out, C, H, W, Ho, Wo, total);
    return lgd::check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of a pointwise convolution from its per-image partial products: out[o][i] = scale[o] * sum_n part[n][o][i]
// (part = bmm(dz (N, Co, HW), x^T (N, HW, Ci)); scale = the frozen per-channel factor of the FrozenBN folded into the filter, or
// null).  torch: a reduce launch + a scale launch per convolution (52 - 200 tiny launches per step).
namespace lgd {
__global__ __launch_bounds__(256) void sum_batch_scale_kernel(const float* __restrict__ part, const float* __restrict__ scale,
                                                              float* __restrict__ out, int N, int Co, int Ci) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, per = (long long)Co * Ci;
    if (i >= per) return;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += part[(long long)n * per + i];
    out[i] = scale ? s * scale[(int)(i / Ci)] : s;
}
}  // namespace lgd

extern "C" int lgd_sum_batch_scale(const float* part, const float* scale, int N, int Co, int Ci, float* out, void* stream) {
    if (!part || !out || N < 1 || Co < 1 || Ci < 1) return LGD_EINVAL;
    const long long per = (long long)Co * Ci;
    LGD_LAUNCH("sum_batch_scale_kernel", lgd::sum_batch_scale_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0,
               (hipStream_t)stream, part, scale, out, N, Co, Ci);
    return lgd::check_launch();
}

namespace lgd {

// out_t[i] = w_t[i] * scale_t[i / cols_t] for a table of tensors in ONE launch: the filter folds w * scale of every trainable 1x1
// convolution that is followed by a FrozenBN (62 of them in R-101), which each cost a 5 us launch per step
// [d2-memory: conv -> FrozenBN of the bottleneck blocks; SURVEY.md appendix A].  Block -> tensor by binary search in the block prefix.
__global__ __launch_bounds__(256) void scale_rows_multi_kernel(const lgd_rows_task* tasks, const int* blk0, int n, unsigned* amax) {
    __shared__ float slots[4];
    int lo = 0, hi = n - 1;
    while (lo < hi) {   // last task whose first block is <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const lgd_rows_task t = tasks[lo];
    const long long total = (long long)t.rows * t.cols;
    const long long i = ((long long)(blockIdx.x - blk0[lo]) * 256 + threadIdx.x) * 4;
    float am = 0.f;
    if (i < total) {
        const float* w = reinterpret_cast<const float*>(t.w);
        const float* sc = reinterpret_cast<const float*>(t.scale);
        float* o = reinterpret_cast<float*>(t.out);
        if ((t.cols & 3) == 0 && ((t.w | t.out) & 15) == 0) {   // a vector lies inside one row
            const float4 v = *reinterpret_cast<const float4*>(w + i);
            const float f = sc[i / t.cols];
            const float4 r = make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
            *reinterpret_cast<float4*>(o + i) = r;
            am = fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w)));
        } else {
            for (long long e = i; e < i + 4 && e < total; ++e) {
                const float r = w[e] * sc[e / t.cols];
                o[e] = r;
                am = fmaxf(am, fabsf(r));
            }
        }
    }
    // max |w * scale| of the task: the f16x2 scale of the filter image (lgd_gemm2h_split) without a pass of its own per filter and step
    if (amax) block_max_bits(amax + lo, wave_max(am), slots);
}

}  // namespace lgd

extern "C" int lgd_scale_rows_multi(const void* tasks_dev, const int32_t* blk0_dev, int n, int nblocks, uint32_t* amax_out, void* stream) {
    if (!tasks_dev || !blk0_dev || n < 1 || nblocks < 1) return LGD_EINVAL;
    LGD_LAUNCH("scale_rows_multi_kernel", lgd::scale_rows_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
               reinterpret_cast<const lgd_rows_task*>(tasks_dev), reinterpret_cast<const int*>(blk0_dev), n, amax_out);
    return lgd::check_launch();
}
