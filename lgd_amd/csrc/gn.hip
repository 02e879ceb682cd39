// K5: GroupNorm(num_groups=1, affine=False, eps=1e-5) [+ReLU] over whole pyramids, and the
// K3b epilogue ReLU(x + ctx[b,c]).
//   [ref: dynamic_teacher/layers.py:6-7,22-32 (get_norm / get_CONVS), dynamic_teacher.py:57,67-73,151]
// torch's native GroupNorm launches one workgroup per (sample, group): with ONE group that is B = 8
// workgroups for a 137 MB tensor (2.8 ms per call at p3 on MI355X, measured) -- 4 GN layers x 5 levels
// forward and backward.  Here a (level, sample) segment is cut into 4096-element chunks, one wave64
// per chunk (dwordx4 loads), fp64 partial moments, a tiny per-segment finalize, then a streaming apply.
// HBM traffic: fwd = read P (stats) + read P + write P (apply); bwd = 2P (stats) + 2P + P (apply).
// The backward recomputes y = (x-mean)*rstd with the forward's exact instruction sequence, so the ReLU
// mask is bit-identical to the forward's without storing it.
#include "common.h"

namespace lgd {

constexpr int kGnChunk = 4096;
constexpr float kGnEps = 1e-5f;

struct GnArgs {
    const float* x[LGD_MAX_LEVELS];
    const float* dy[LGD_MAX_LEVELS];
    float* out[LGD_MAX_LEVELS];   // fwd: y ; bwd: dx
    int N[LGD_MAX_LEVELS];        // elements per sample = C*H*W
    int cps[LGD_MAX_LEVELS];      // chunks per sample
    int wave0[LGD_MAX_LEVELS + 1];
    int L, B, relu, nwaves;
    double* ws;                   // [nwaves][2]
    float* stats;                 // fwd: [L*B][2] mean,rstd ; bwd also reads it
    float* bstats;                // bwd: [L*B][2] m1,m2
};

struct GnWhere { int l, b, chunk, seg; };
__device__ __forceinline__ GnWhere gn_locate(const GnArgs& a, int w) {
    int l = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) l += (i < a.L && w >= a.wave0[i]) ? 1 : 0;
    const int local = w - a.wave0[l];
    GnWhere r;
    r.l = l; r.b = local / a.cps[l]; r.chunk = local % a.cps[l]; r.seg = l * a.B + r.b;
    return r;
}

__device__ __forceinline__ float gn_norm(float x, float mu, float r) { return __fmul_rn(__fsub_rn(x, mu), r); }

template <int MODE>  // 0: fwd stats (x, x^2)   1: bwd stats (g, g*xhat)
__global__ __launch_bounds__(256) void gn_stats_kernel(GnArgs a) {
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= a.nwaves) return;
    const int lane = threadIdx.x & 63;
    const GnWhere q = gn_locate(a, w);
    const int N = a.N[q.l];
    const float* __restrict__ px = a.x[q.l] + (size_t)q.b * N;
    const float* __restrict__ pd = MODE == 1 ? a.dy[q.l] + (size_t)q.b * N : nullptr;
    float mu = 0.f, r = 1.f;
    if (MODE == 1) { mu = a.stats[2 * q.seg]; r = a.stats[2 * q.seg + 1]; }
    const int e0 = q.chunk * kGnChunk, e1 = min(N, e0 + kGnChunk);
    double s0 = 0, s1 = 0;
    for (int e = e0 + lane * 4; e < e1; e += 1024) {
        float4 vx[4], vd[4];
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ee = e + u * 256;
            if (ee < e1) {
                vx[u] = ldg_stream4(px + ee);
                if (MODE == 1) vd[u] = ldg_stream4(pd + ee);
            } else { vx[u] = make_float4(0, 0, 0, 0); vd[u] = make_float4(0, 0, 0, 0); }
        }
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xs[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
            const float ds[4] = {vd[u].x, vd[u].y, vd[u].z, vd[u].w};
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 0) {
                    const double x = xs[j];
                    s0 += x; s1 = fma(x, x, s1);
                } else {
                    const float xh = gn_norm(xs[j], mu, r);
                    const float g = (a.relu && !(xh > 0.f)) ? 0.f : ds[j];
                    s0 += (double)g; s1 = fma((double)g, (double)xh, s1);
                }
            }
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (lane == 0) { a.ws[2 * (size_t)w] = s0; a.ws[2 * (size_t)w + 1] = s1; }
}

template <int MODE>
__global__ __launch_bounds__(256) void gn_finalize_kernel(GnArgs a) {
    __shared__ double red[8];
    const int seg = blockIdx.x, l = seg / a.B, b = seg % a.B;
    const int cps = a.cps[l];
    const double* p = a.ws + 2 * ((size_t)a.wave0[l] + (size_t)b * cps);
    double s0 = 0, s1 = 0;
    for (int c = threadIdx.x; c < cps; c += 256) { s0 += p[2 * c]; s1 += p[2 * c + 1]; }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = s0; red[2 * (threadIdx.x >> 6) + 1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s0 = (red[0] + red[2]) + (red[4] + red[6]);
        s1 = (red[1] + red[3]) + (red[5] + red[7]);
        const double n = (double)a.N[l];
        if (MODE == 0) {
            const double m = s0 / n, var = fmax(s1 / n - m * m, 0.0);
            a.stats[2 * seg] = (float)m;
            a.stats[2 * seg + 1] = (float)(1.0 / sqrt(var + (double)kGnEps));
        } else {
            a.bstats[2 * seg] = (float)(s0 / n);
            a.bstats[2 * seg + 1] = (float)(s1 / n);
        }
    }
}

template <int MODE>  // 0: y = relu?((x-mu)*r)   1: dx = r*(g - m1 - xhat*m2)
__global__ __launch_bounds__(256) void gn_apply_kernel(GnArgs a) {
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= a.nwaves) return;
    const int lane = threadIdx.x & 63;
    const GnWhere q = gn_locate(a, w);
    const int N = a.N[q.l];
    const float* __restrict__ px = a.x[q.l] + (size_t)q.b * N;
    const float* __restrict__ pd = MODE == 1 ? a.dy[q.l] + (size_t)q.b * N : nullptr;
    float* __restrict__ po = a.out[q.l] + (size_t)q.b * N;
    const float mu = a.stats[2 * q.seg], r = a.stats[2 * q.seg + 1];
    float m1 = 0.f, m2 = 0.f;
    if (MODE == 1) { m1 = a.bstats[2 * q.seg]; m2 = a.bstats[2 * q.seg + 1]; }
    const int e0 = q.chunk * kGnChunk, e1 = min(N, e0 + kGnChunk);
    for (int e = e0 + lane * 4; e < e1; e += 1024) {
        float4 vx[4], vd[4];
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ee = e + u * 256;
            if (ee < e1) {
                vx[u] = ldg_stream4(px + ee);
                if (MODE == 1) vd[u] = ldg_stream4(pd + ee);
            }
        }
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ee = e + u * 256;
            if (ee >= e1) continue;
            const float xs[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
            const float ds[4] = {vd[u].x, vd[u].y, vd[u].z, vd[u].w};
            float o[4];
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = gn_norm(xs[j], mu, r);
                if (MODE == 0) {
                    o[j] = a.relu ? fmaxf(xh, 0.f) : xh;
                } else {
                    const float g = (a.relu && !(xh > 0.f)) ? 0.f : ds[j];
                    o[j] = r * (g - m1 - xh * m2);
                }
            }
            *reinterpret_cast<float4*>(po + ee) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ---------------------------------------------------------------- ReLU(x + ctx[b,c]) and its backward
struct CtxArgs {
    const float* x[LGD_MAX_LEVELS];   // fwd: conv output ; bwd: y (saved output)
    const float* dy[LGD_MAX_LEVELS];
    float* out[LGD_MAX_LEVELS];       // fwd: y ; bwd: dx
    int HW[LGD_MAX_LEVELS];
    int blk0[LGD_MAX_LEVELS + 1];
    const float* ctx;                 // fwd: [L][B][C]
    float* dctx;                      // bwd: [L][B][C]
    int L, BC;
};

// one wave per (level, b, c) plane; 4 planes per workgroup
template <int MODE>
__global__ __launch_bounds__(256) void ctx_relu_kernel(CtxArgs a) {
    int l = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) l += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    const int plane = ((int)blockIdx.x - a.blk0[l]) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int HW = a.HW[l];
    const float* __restrict__ px = a.x[l] + (size_t)plane * HW;
    const float* __restrict__ pd = MODE == 1 ? a.dy[l] + (size_t)plane * HW : nullptr;
    float* __restrict__ po = a.out[l] + (size_t)plane * HW;
    const float c = MODE == 0 ? a.ctx[(size_t)l * a.BC + plane] : 0.f;
    float acc = 0.f;
    if ((HW & 3) == 0) {
        for (int e = lane * 4; e < HW; e += 256 * 4) {  // 4 independent 16-byte loads per tensor in flight
            float4 v[4], d[4];
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * 256;
                if (ee < HW) { v[u] = ldg_stream4(px + ee); if (MODE == 1) d[u] = ldg_stream4(pd + ee); }
            }
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * 256;
                if (ee >= HW) continue;
                float4 o;
                if (MODE == 0) {
                    o = make_float4(fmaxf(v[u].x + c, 0.f), fmaxf(v[u].y + c, 0.f), fmaxf(v[u].z + c, 0.f), fmaxf(v[u].w + c, 0.f));
                } else {
                    o = make_float4(v[u].x > 0.f ? d[u].x : 0.f, v[u].y > 0.f ? d[u].y : 0.f, v[u].z > 0.f ? d[u].z : 0.f,
                                    v[u].w > 0.f ? d[u].w : 0.f);
                    acc += (o.x + o.y) + (o.z + o.w);
                }
                *reinterpret_cast<float4*>(po + ee) = o;
            }
        }
    } else {
        for (int e = lane; e < HW; e += 64 * 8) {
            float v[8], d[8];
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ee = e + u * 64;
                if (ee < HW) { v[u] = ldg_stream(px + ee); if (MODE == 1) d[u] = ldg_stream(pd + ee); }
            }
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ee = e + u * 64;
                if (ee >= HW) continue;
                if (MODE == 0) { po[ee] = fmaxf(v[u] + c, 0.f); }
                else { const float o = v[u] > 0.f ? d[u] : 0.f; po[ee] = o; acc += o; }
            }
        }
    }
    if (MODE == 1) {
        acc = wave_sum(acc);
        if (lane == 0) a.dctx[(size_t)l * a.BC + plane] = acc;
    }
}

static int gn_fill(GnArgs& a, const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C, int relu) {
    if (!x_host || !level_hw_host || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 4 || (C & 3)) return LGD_EINVAL;
    a.L = L; a.B = B; a.relu = relu;
    int w = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.x[l] = nullptr; a.dy[l] = nullptr; a.out[l] = nullptr;
        a.wave0[l] = w;
        if (l < L) {
            if (!x_host[l]) return LGD_EINVAL;
            a.x[l] = x_host[l];
            a.N[l] = C * level_hw_host[2 * l] * level_hw_host[2 * l + 1];
            a.cps[l] = (a.N[l] + kGnChunk - 1) / kGnChunk;
            w += B * a.cps[l];
        } else { a.N[l] = 0; a.cps[l] = 1; }
    }
    a.wave0[LGD_MAX_LEVELS] = w;
    a.nwaves = w;
    return LGD_OK;
}

}  // namespace lgd

extern "C" {

size_t lgd_gn1_ws_doubles(const int32_t* level_hw_host, int L, int B, int C) {
    size_t w = 0;
    for (int l = 0; l < L; ++l) {
        const int n = C * level_hw_host[2 * l] * level_hw_host[2 * l + 1];
        w += (size_t)B * ((n + lgd::kGnChunk - 1) / lgd::kGnChunk);
    }
    return 2 * w;
}

int lgd_gn1_fwd(const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C, int relu, double* ws,
                float* stats, float* const* y_host, void* stream) {
    lgd::GnArgs a;
    if (lgd::gn_fill(a, x_host, level_hw_host, L, B, C, relu) != LGD_OK || !ws || !stats || !y_host) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!y_host[l]) return LGD_EINVAL; a.out[l] = y_host[l]; }
    a.ws = ws; a.stats = stats; a.bstats = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((a.nwaves + 3) / 4);
    LGD_LAUNCH("gn_stats_kernel", lgd::gn_stats_kernel<0>, grid, dim3(256), 0, s, a);
    LGD_LAUNCH("gn_finalize_kernel", lgd::gn_finalize_kernel<0>, dim3(L * B), dim3(256), 0, s, a);
    LGD_LAUNCH("gn_apply_kernel", lgd::gn_apply_kernel<0>, grid, dim3(256), 0, s, a);
    return lgd::check_launch();
}

int lgd_gn1_stats(const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C, double* ws, float* stats,
                  void* stream) {
    lgd::GnArgs a;
    if (lgd::gn_fill(a, x_host, level_hw_host, L, B, C, 0) != LGD_OK || !ws || !stats) return LGD_EINVAL;
    a.ws = ws; a.stats = stats; a.bstats = nullptr;
    hipStream_t s = (hipStream_t)stream;
    LGD_LAUNCH("gn_stats_kernel", lgd::gn_stats_kernel<0>, dim3((a.nwaves + 3) / 4), dim3(256), 0, s, a);
    LGD_LAUNCH("gn_finalize_kernel", lgd::gn_finalize_kernel<0>, dim3(L * B), dim3(256), 0, s, a);
    return lgd::check_launch();
}

int lgd_gn1_bwd(const float* const* x_host, const float* const* dy_host, const int32_t* level_hw_host, int L, int B, int C,
                int relu, const float* stats, double* ws, float* bstats, float* const* dx_host, void* stream) {
    lgd::GnArgs a;
    if (lgd::gn_fill(a, x_host, level_hw_host, L, B, C, relu) != LGD_OK || !dy_host || !stats || !ws || !bstats || !dx_host)
        return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!dy_host[l] || !dx_host[l]) return LGD_EINVAL;
        a.dy[l] = dy_host[l]; a.out[l] = dx_host[l];
    }
    a.ws = ws; a.stats = const_cast<float*>(stats); a.bstats = bstats;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((a.nwaves + 3) / 4);
    LGD_LAUNCH("gn_bwd_stats_kernel", lgd::gn_stats_kernel<1>, grid, dim3(256), 0, s, a);
    LGD_LAUNCH("gn_bwd_finalize_kernel", lgd::gn_finalize_kernel<1>, dim3(L * B), dim3(256), 0, s, a);
    LGD_LAUNCH("gn_bwd_apply_kernel", lgd::gn_apply_kernel<1>, grid, dim3(256), 0, s, a);
    return lgd::check_launch();
}

static int ctx_fill(lgd::CtxArgs& a, const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C) {
    if (!x_host || !level_hw_host || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 4 || (C & 3)) return LGD_EINVAL;
    a.L = L; a.BC = B * C;
    int blk = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.x[l] = nullptr; a.dy[l] = nullptr; a.out[l] = nullptr;
        a.blk0[l] = blk;
        a.HW[l] = l < L ? level_hw_host[2 * l] * level_hw_host[2 * l + 1] : 0;
        if (l < L) { if (!x_host[l]) return LGD_EINVAL; a.x[l] = x_host[l]; blk += B * C / 4; }
    }
    a.blk0[LGD_MAX_LEVELS] = blk;
    return blk;
}

int lgd_ctx_relu_fwd(const float* const* x_host, const float* ctx, const int32_t* level_hw_host, int L, int B, int C,
                     float* const* y_host, void* stream) {
    lgd::CtxArgs a;
    const int nblk = ctx_fill(a, x_host, level_hw_host, L, B, C);
    if (nblk < 0 || !ctx || !y_host) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!y_host[l]) return LGD_EINVAL; a.out[l] = y_host[l]; }
    a.ctx = ctx; a.dctx = nullptr;
    LGD_LAUNCH("ctx_relu_kernel", lgd::ctx_relu_kernel<0>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

int lgd_ctx_relu_bwd(const float* const* y_host, const float* const* dy_host, const int32_t* level_hw_host, int L, int B,
                     int C, float* const* dx_host, float* dctx, void* stream) {
    lgd::CtxArgs a;
    const int nblk = ctx_fill(a, y_host, level_hw_host, L, B, C);
    if (nblk < 0 || !dy_host || !dx_host || !dctx) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!dy_host[l] || !dx_host[l]) return LGD_EINVAL;
        a.dy[l] = dy_host[l]; a.out[l] = dx_host[l];
    }
    a.ctx = nullptr; a.dctx = dctx;
    LGD_LAUNCH("ctx_relu_bwd_kernel", lgd::ctx_relu_kernel<1>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
