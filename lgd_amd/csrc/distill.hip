// K4: feature-distillation loss = coef * mean((InstanceNorm(a) - InstanceNorm(b))^2) over all levels.
//   [ref: models/base_distillator.py:59-64]
// The reference runs 10 InstanceNorm kernels, 10 views, 2 concatenations (copies of both pyramids)
// and an MSE reduction: ~8 passes over 2 pyramids.  Here: ONE pass.  Per (level,b,c) plane the five
// moments  Sa, Saa, Sb, Sbb, Sab  are accumulated in fp64 and the plane's contribution follows in
// closed form:
//   sum (a^ - b^)^2 = N * [ va*ra^2 + vb*rb^2 - 2*cov*ra*rb ],   ra = 1/sqrt(va+eps), rb likewise
// (biased variances, eps = 1e-5 inside the sqrt, exactly InstanceNorm2d's definition).
// Algorithmic HBM traffic: 2 pyramids read once (fwd); bwd reads both and writes one.
//
// Mapping: a wave64 owns one 4096-element chunk of one plane (16 x dwordx4 per lane per tensor,
// fully coalesced); 4 waves per workgroup; partial moments go to a fp64 workspace, a second
// tiny kernel folds them per plane (fixed order => deterministic), a third sums the per-block terms.
#include "common.h"

namespace lgd {

constexpr int kChunk = 4096;
constexpr float kEps = 1e-5f;

struct DistArgs {
    const float* a[LGD_MAX_LEVELS];
    const float* b[LGD_MAX_LEVELS];
    float* ga[LGD_MAX_LEVELS];
    int HW[LGD_MAX_LEVELS], cpp[LGD_MAX_LEVELS];  // chunks per plane
    int wave0[LGD_MAX_LEVELS + 1];                // first wave (chunk) of each level
    int plane0[LGD_MAX_LEVELS + 1];               // first plane of each level
    int L, BC;
    double* ws;          // [nwaves][5] partial moments | [nfin_blocks] block terms
    float* stats;        // [nplanes][8]
    float* loss;
    const float* grad_loss;
    float coef;
    double inv_total;    // 1 / (B*C*sum HW)
    int nwaves, nplanes, nfin;
};

struct Where { int l, plane_local, chunk, gplane; };

__device__ __forceinline__ Where locate_wave(const DistArgs& a, int w) {
    int l = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) l += (i < a.L && w >= a.wave0[i]) ? 1 : 0;
    const int local = w - a.wave0[l];
    Where r;
    r.l = l; r.plane_local = local / a.cpp[l]; r.chunk = local % a.cpp[l];
    r.gplane = a.plane0[l] + r.plane_local;
    return r;
}

__global__ __launch_bounds__(256) void in_moments_kernel(DistArgs a) {
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= a.nwaves) return;
    const int lane = threadIdx.x & 63;
    const Where q = locate_wave(a, w);
    const int HW = a.HW[q.l];
    const size_t base = (size_t)q.plane_local * HW;
    const float* __restrict__ pa = a.a[q.l] + base;
    const float* __restrict__ pb = a.b[q.l] + base;
    const int e0 = q.chunk * kChunk, e1 = min(HW, e0 + kChunk);
    double sa = 0, saa = 0, sb = 0, sbb = 0, sab = 0;
    if ((HW & 3) == 0) {
        for (int e = e0 + lane * 4; e < e1; e += 256 * 4) {  // 4 independent 16-byte loads per tensor in flight
            float4 va[4], vb[4];
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * 256;
                if (ee < e1) { va[u] = ldg_stream4(pa + ee); vb[u] = ldg_stream4(pb + ee); }
                else { va[u] = make_float4(0, 0, 0, 0); vb[u] = make_float4(0, 0, 0, 0); }
            }
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xa[4] = {va[u].x, va[u].y, va[u].z, va[u].w};
                const float xb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double x = xa[j], y = xb[j];
                    sa += x; saa = fma(x, x, saa); sb += y; sbb = fma(y, y, sbb); sab = fma(x, y, sab);
                }
            }
        }
    } else {
        // planes whose size is not a multiple of 4 floats (p5..p7 at most sizes): 8 independent loads per tensor in
        // flight per lane -- a dependent one-load-per-iteration loop here was the tail of the whole launch
        for (int e = e0 + lane; e < e1; e += 64 * 8) {
            float xa[8], xb[8];
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ee = e + u * 64;
                xa[u] = ee < e1 ? ldg_stream(pa + ee) : 0.f;
                xb[u] = ee < e1 ? ldg_stream(pb + ee) : 0.f;
            }
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double x = xa[u], y = xb[u];
                sa += x; saa = fma(x, x, saa); sb += y; sbb = fma(y, y, sbb); sab = fma(x, y, sab);
            }
        }
    }
    sa = wave_sum(sa); saa = wave_sum(saa); sb = wave_sum(sb); sbb = wave_sum(sbb); sab = wave_sum(sab);
    if (lane == 0) {
        double* o = a.ws + (size_t)w * 5;
        o[0] = sa; o[1] = saa; o[2] = sb; o[3] = sbb; o[4] = sab;
    }
}

__global__ __launch_bounds__(256) void in_finalize_kernel(DistArgs a) {
    __shared__ double red[4];
    const int gp = blockIdx.x * 256 + threadIdx.x;
    double term = 0.0;
    if (gp < a.nplanes) {
        int l = 0;
        #pragma unroll
        for (int i = 1; i < LGD_MAX_LEVELS; ++i) l += (i < a.L && gp >= a.plane0[i]) ? 1 : 0;
        const int pl = gp - a.plane0[l];
        const int cpp = a.cpp[l];
        const double* p = a.ws + ((size_t)a.wave0[l] + (size_t)pl * cpp) * 5;
        double sa = 0, saa = 0, sb = 0, sbb = 0, sab = 0;
        for (int c = 0; c < cpp; ++c) { sa += p[0]; saa += p[1]; sb += p[2]; sbb += p[3]; sab += p[4]; p += 5; }
        const double n = (double)a.HW[l];
        const double ma = sa / n, mb = sb / n;
        const double va = fmax(saa / n - ma * ma, 0.0), vb = fmax(sbb / n - mb * mb, 0.0);
        const double cov = sab / n - ma * mb;
        const double ra = 1.0 / sqrt(va + (double)kEps), rb = 1.0 / sqrt(vb + (double)kEps);
        term = n * (va * ra * ra + vb * rb * rb - 2.0 * cov * ra * rb);
        float* s = a.stats + (size_t)gp * 8;
        s[0] = (float)ma; s[1] = (float)ra; s[2] = (float)mb; s[3] = (float)rb;
        s[4] = (float)(va * ra * ra - cov * ra * rb);  // q = mean(d * a^), d = a^ - b^
        s[5] = 0.f; s[6] = 0.f; s[7] = 0.f;
    }
    term = wave_sum(term);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) a.ws[(size_t)a.nwaves * 5 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void in_loss_kernel(DistArgs a) {
    const double* p = a.ws + (size_t)a.nwaves * 5;
    double s = 0.0;
    for (int i = threadIdx.x; i < a.nfin; i += 64) s += p[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) a.loss[0] = (float)((double)a.coef * s * a.inv_total);
}

// d loss / d a  for one chunk:  g * 2*coef/Ntot * ra * (a^*(1-q) - b^)
__global__ __launch_bounds__(256) void in_mse_bwd_kernel(DistArgs a) {
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= a.nwaves) return;
    const int lane = threadIdx.x & 63;
    const Where q = locate_wave(a, w);
    const int HW = a.HW[q.l];
    const size_t base = (size_t)q.plane_local * HW;
    const float* __restrict__ pa = a.a[q.l] + base;
    const float* __restrict__ pb = a.b[q.l] + base;
    float* __restrict__ pg = a.ga[q.l] + base;
    const float* s = a.stats + (size_t)q.gplane * 8;
    const float ma = s[0], ra = s[1], mb = s[2], rb = s[3], qq = s[4];
    const float g = a.grad_loss[0] * (float)(2.0 * (double)a.coef * a.inv_total) * ra;
    const float ka = ra * (1.f - qq);
    const int e0 = q.chunk * kChunk, e1 = min(HW, e0 + kChunk);
    if ((HW & 3) == 0) {
        for (int e = e0 + lane * 4; e < e1; e += 256 * 4) {
            float4 va[4], vb[4];
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * 256;
                if (ee < e1) { va[u] = ldg_stream4(pa + ee); vb[u] = ldg_stream4(pb + ee); }
            }
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * 256;
                if (ee < e1) {
                    float4 o;
                    o.x = g * ((va[u].x - ma) * ka - (vb[u].x - mb) * rb);
                    o.y = g * ((va[u].y - ma) * ka - (vb[u].y - mb) * rb);
                    o.z = g * ((va[u].z - ma) * ka - (vb[u].z - mb) * rb);
                    o.w = g * ((va[u].w - ma) * ka - (vb[u].w - mb) * rb);
                    *reinterpret_cast<float4*>(pg + ee) = o;
                }
            }
        }
    } else {
        for (int e = e0 + lane; e < e1; e += 64 * 8) {
            float xa[8], xb[8];
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ee = e + u * 64;
                if (ee < e1) { xa[u] = ldg_stream(pa + ee); xb[u] = ldg_stream(pb + ee); }
            }
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ee = e + u * 64;
                if (ee < e1) pg[ee] = g * ((xa[u] - ma) * ka - (xb[u] - mb) * rb);
            }
        }
    }
}

static int fill(DistArgs& a, const float* const* a_host, const float* const* b_host, const int32_t* level_hw_host,
                int L, int B, int C, float coef) {
    if (!a_host || !b_host || !level_hw_host || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 1) return LGD_EINVAL;
    a.L = L; a.BC = B * C; a.coef = coef;
    int w = 0, pl = 0;
    double tot = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.a[l] = a.b[l] = nullptr; a.ga[l] = nullptr;
        a.wave0[l] = w; a.plane0[l] = pl;
        if (l < L) {
            if (!a_host[l] || !b_host[l]) return LGD_EINVAL;
            a.a[l] = a_host[l]; a.b[l] = b_host[l];
            a.HW[l] = level_hw_host[2 * l] * level_hw_host[2 * l + 1];
            a.cpp[l] = (a.HW[l] + kChunk - 1) / kChunk;
            w += B * C * a.cpp[l]; pl += B * C;
            tot += (double)B * C * a.HW[l];
        } else { a.HW[l] = 0; a.cpp[l] = 1; }
    }
    a.wave0[LGD_MAX_LEVELS] = w; a.plane0[LGD_MAX_LEVELS] = pl;
    a.nwaves = w; a.nplanes = pl; a.nfin = (pl + 255) / 256;
    a.inv_total = 1.0 / tot;
    return LGD_OK;
}

}  // namespace lgd

extern "C" {

size_t lgd_distill_ws_doubles(const int32_t* level_hw_host, int L, int B, int C) {
    size_t w = 0, pl = 0;
    for (int l = 0; l < L; ++l) {
        const int hw = level_hw_host[2 * l] * level_hw_host[2 * l + 1];
        w += (size_t)B * C * ((hw + lgd::kChunk - 1) / lgd::kChunk);
        pl += (size_t)B * C;
    }
    return w * 5 + (pl + 255) / 256;
}

int lgd_distill_fwd(const float* const* a_host, const float* const* b_host, const int32_t* level_hw_host, int L, int B,
                    int C, float coef, double* ws, float* stats, float* loss, void* stream) {
    lgd::DistArgs a;
    if (lgd::fill(a, a_host, b_host, level_hw_host, L, B, C, coef) != LGD_OK || !ws || !stats || !loss) return LGD_EINVAL;
    a.ws = ws; a.stats = stats; a.loss = loss; a.grad_loss = nullptr;
    hipStream_t s = (hipStream_t)stream;
    LGD_LAUNCH("in_moments_kernel", lgd::in_moments_kernel, dim3((a.nwaves + 3) / 4), dim3(256), 0, s, a);
    LGD_LAUNCH("in_finalize_kernel", lgd::in_finalize_kernel, dim3(a.nfin), dim3(256), 0, s, a);
    LGD_LAUNCH("in_loss_kernel", lgd::in_loss_kernel, dim3(1), dim3(64), 0, s, a);
    return lgd::check_launch();
}

int lgd_distill_bwd(const float* const* a_host, const float* const* b_host, const int32_t* level_hw_host, int L, int B,
                    int C, float coef, const float* stats, const float* grad_loss, float* const* grad_a_host,
                    void* stream) {
    lgd::DistArgs a;
    if (lgd::fill(a, a_host, b_host, level_hw_host, L, B, C, coef) != LGD_OK || !stats || !grad_loss || !grad_a_host)
        return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!grad_a_host[l]) return LGD_EINVAL; a.ga[l] = grad_a_host[l]; }
    a.ws = nullptr; a.stats = const_cast<float*>(stats); a.loss = nullptr; a.grad_loss = grad_loss;
    LGD_LAUNCH("in_mse_bwd_kernel", lgd::in_mse_bwd_kernel, dim3((a.nwaves + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
