// GroupNorm(G groups, per-channel affine, eps 1e-5) [+ReLU] over a list of maps in one call: the FCOS towers'
// conv3x3 -> GroupNorm(32, 256) -> ReLU  [ref: models/customized_detectors/thirdparty_heads/fcos.py:455-470, 520-531],
// one module shared by all pyramid levels (and, with the single head pass over student + teacher features, by both
// pyramids).  torch runs this as native_group_norm + relu per level: 2 x 10 launches per tower layer and three passes over
// the activations; here it is stats (read P) + apply (read P, write P) for ALL maps, and the backward recomputes the
// normalised value with the forward's exact instruction sequence so the ReLU mask needs no storage.
// A group is C/G adjacent channel planes of one sample = one contiguous run of (C/G)*HW floats; a wave owns one
// <= 4096-element chunk of ONE channel plane, so gamma/beta are wave-uniform scalars.
//   backward: g = dy * [y > 0];  dxhat = gamma_c * g;  dx = rstd * (dxhat - mean_grp(dxhat) - xhat * mean_grp(dxhat * xhat))
//             dgamma_c = sum_{maps, b, hw} g * xhat;  dbeta_c = sum g   (per-plane sums are emitted, the host adds the planes)
#include "common.h"

// one rounding per source operation: the backward must reproduce the forward's y bit for bit (ReLU mask by recomputation)
#pragma clang fp contract(off)

namespace lgd {

constexpr int kGgChunk = 4096;
constexpr float kGgEps = 1e-5f;

struct GgArgs {
    const float* x[LGD_MAX_LEVELS];
    const float* dy[LGD_MAX_LEVELS];
    float* out[LGD_MAX_LEVELS];     // fwd: y ; bwd: dx
    int HW[LGD_MAX_LEVELS];
    int cpp[LGD_MAX_LEVELS];        // chunks per channel plane
    int chunk[LGD_MAX_LEVELS];      // elements per chunk (balanced, multiple of 4)
    int wave0[LGD_MAX_LEVELS + 1];  // first wave of the level; level l has B*C*cpp[l] waves, plane-major
    int L, B, C, G, relu, nwaves;
    const float* gamma;             // [C] or null (= 1)
    const float* beta;              // [C] or null (= 0)
    double* ws;                     // [nwaves][2]
    float* stats;                   // [L*B*G][2] mean, rstd
    float* bstats;                  // [L*B*G][2] m1, m2 (backward)
    float* plane_sums;              // [L*B*C][2] sum g, sum g*xhat (backward)
    float* affine;                  // optional (forward finalize): [L*B][C][2] rstd*gamma, beta - mean*rstd*gamma
    float* wmax;                    // optional (backward, with coef): [nwaves][2] max |g|, max |xhat| of every chunk -- the finalize turns them into
    unsigned* bound_out;            //   max |dx| over everything (float bits, atomic max into a word the caller zeroed): the f16x2 scale of lgd_wino_out_t_gn_h2
    float* coef;                    // optional (backward finalize): [L*B][C][4] ca = rstd*gamma, cm = rstd*m1, mean, cb = rstd^2*m2 --
                                    //   dx = ca*g - cm - (x - mean)*cb, applied by the producing convolution's lgd_wino_out_t_gn
};

struct GgWhere { int l, plane, chunk; };
__device__ __forceinline__ GgWhere gg_locate(const GgArgs& a, int w) {
    int l = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) l += (i < a.L && w >= a.wave0[i]) ? 1 : 0;
    const int local = w - a.wave0[l];
    GgWhere r;
    r.l = l; r.plane = local / a.cpp[l]; r.chunk = local % a.cpp[l];
    return r;
}

__device__ __forceinline__ float gg_hat(float x, float mu, float r) { return __fmul_rn(__fsub_rn(x, mu), r); }
__device__ __forceinline__ float gg_affine(float xh, float ga, float be) { return __fadd_rn(__fmul_rn(xh, ga), be); }

// MODE 0: (sum x, sum x^2); MODE 1: (sum g, sum g*xhat), g = dy masked by the recomputed ReLU
template <int MODE>
__global__ __launch_bounds__(256) void gg_stats_kernel(GgArgs a) {
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= a.nwaves) return;
    const int lane = threadIdx.x & 63;
    const GgWhere q = gg_locate(a, w);
    const int HW = a.HW[q.l];
    const float* __restrict__ px = a.x[q.l] + (size_t)q.plane * HW;
    const float* __restrict__ pd = MODE == 1 ? a.dy[q.l] + (size_t)q.plane * HW : nullptr;
    const int b = q.plane / a.C, c = q.plane % a.C, cg = a.C / a.G;
    const int seg = (q.l * a.B + b) * a.G + c / cg;
    float mu = 0.f, r = 1.f, ga = 1.f, be = 0.f;
    if (MODE == 1) {
        mu = a.stats[2 * seg]; r = a.stats[2 * seg + 1];
        if (a.gamma) ga = a.gamma[c];
        if (a.beta) be = a.beta[c];
    }
    const int e0 = q.chunk * a.chunk[q.l], e1 = min(HW, e0 + a.chunk[q.l]);
    double s0 = 0, s1 = 0;
    float mg = 0.f, mx = 0.f;
    auto acc = [&](float x, float d) {
        if (MODE == 0) { const double xd = x; s0 += xd; s1 = fma(xd, xd, s1); }
        else {
            const float xh = gg_hat(x, mu, r);
            const float g = (a.relu && !(gg_affine(xh, ga, be) > 0.f)) ? 0.f : d;
            s0 += (double)g; s1 = fma((double)g, (double)xh, s1);
            mg = fmaxf(mg, fabsf(g)); mx = fmaxf(mx, fabsf(xh));
        }
    };
    if ((HW & 3) == 0) {
        for (int e = e0 + lane * 4; e < e1; e += 1024) {
            float4 vx[4], vd[4];
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * 256;
                vx[u] = vd[u] = make_float4(0, 0, 0, 0);
                if (ee < e1) { vx[u] = ldg_stream4(px + ee); if (MODE == 1) vd[u] = ldg_stream4(pd + ee); }
            }
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (e + u * 256 >= e1) continue;
                acc(vx[u].x, vd[u].x); acc(vx[u].y, vd[u].y); acc(vx[u].z, vd[u].z); acc(vx[u].w, vd[u].w);
            }
        }
    } else {
        for (int e = e0 + lane; e < e1; e += 64) acc(px[e], MODE == 1 ? pd[e] : 0.f);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (lane == 0) { a.ws[2 * (size_t)w] = s0; a.ws[2 * (size_t)w + 1] = s1; }
    if (MODE == 1 && a.wmax) {
        mg = wave_max(mg); mx = wave_max(mx);
        if (lane == 0) { a.wmax[2 * (size_t)w] = mg; a.wmax[2 * (size_t)w + 1] = mx; }
    }
}

// one workgroup per (level, sample, group): its C/G planes x cpp chunks are contiguous in ws.  A THREAD owns a channel plane (its cpp
// chunk partials summed in order: the plane sums the backward emits), the planes are combined by a fixed-order wave / workgroup
// reduction, and every thread writes its own channel's folded coefficients.  (Round 3's form -- one wave walking the group's planes
// one after the other, a wave-wide fp64 reduction per plane -- took 155 us for GroupNorm(1)'s 256-plane groups against 6 us for
// GroupNorm(32)'s 8-plane ones; this one serves both.)
template <int MODE>
__global__ __launch_bounds__(256) void gg_finalize_kernel(GgArgs a) {
    __shared__ double red[4][2];
    __shared__ float bc[2];
    const int seg = blockIdx.x, g = seg % a.G, lb = seg / a.G, l = lb / a.B, b = lb % a.B;
    const int cg = a.C / a.G, cpp = a.cpp[l];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const double n = (double)cg * (double)a.HW[l];
    double t0 = 0, t1 = 0;
    for (int j = t; j < cg; j += 256) {
        const int c = g * cg + j, plane = b * a.C + c;
        const double* p = a.ws + 2 * ((size_t)a.wave0[l] + (size_t)plane * cpp);
        double s0 = 0, s1 = 0;
        for (int k = 0; k < cpp; ++k) { s0 += p[2 * k]; s1 += p[2 * k + 1]; }
        if (MODE == 0) { t0 += s0; t1 += s1; }
        else {
            const double ga = a.gamma ? (double)a.gamma[c] : 1.0;
            t0 += ga * s0; t1 += ga * s1;
            a.plane_sums[2 * ((size_t)l * a.B * a.C + plane)] = (float)s0;
            a.plane_sums[2 * ((size_t)l * a.B * a.C + plane) + 1] = (float)s1;
        }
    }
    t0 = wave_sum(t0); t1 = wave_sum(t1);
    if (lane == 0) { red[wv][0] = t0; red[wv][1] = t1; }
    __syncthreads();
    if (t == 0) {
        const double u0 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]), u1 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        if (MODE == 0) {
            const double m = u0 / n, var = fmax(u1 / n - m * m, 0.0);
            bc[0] = (float)m; bc[1] = (float)(1.0 / sqrt(var + (double)kGgEps));
            a.stats[2 * seg] = bc[0];
            a.stats[2 * seg + 1] = bc[1];
        } else {
            bc[0] = (float)(u0 / n); bc[1] = (float)(u1 / n);
            a.bstats[2 * seg] = bc[0];
            a.bstats[2 * seg + 1] = bc[1];
        }
    }
    __syncthreads();
    if (MODE == 0 && a.affine) {   // y = gamma * (x - mean) * rstd + beta as x * scale + shift: what the next convolution's input transform applies
        const float mf = bc[0], rf = bc[1];
        for (int j = t; j < cg; j += 256) {
            const int c = g * cg + j;
            const float sc = rf * (a.gamma ? a.gamma[c] : 1.f);
            a.affine[2 * ((size_t)lb * a.C + c)] = sc;
            a.affine[2 * ((size_t)lb * a.C + c) + 1] = (a.beta ? a.beta[c] : 0.f) - mf * sc;
        }
    }
    if (MODE == 1 && a.coef) {
        const float m1 = bc[0], m2 = bc[1];
        const float mu = a.stats[2 * seg], r = a.stats[2 * seg + 1];
        for (int j = t; j < cg; j += 256) {
            const int c = g * cg + j;
            float* k = a.coef + 4 * ((size_t)lb * a.C + c);
            k[0] = r * (a.gamma ? a.gamma[c] : 1.f); k[1] = r * m1; k[2] = mu; k[3] = r * r * m2;
        }
        if (a.wmax && a.bound_out) {
            // |dx| = |ca g - cm - (x - mean) cb| <= |ca| max|g| + |cm| + |r m2| max|xhat| PER PLANE (x - mean = xhat / r): the maxima of one plane
            // against the coefficients of the same plane -- a global max|g| max|ca| multiplies maxima of different channels (a channel group
            // with a tiny variance has a huge rstd and tiny gradients) and wastes the f16 pair's precision window
            float bnd = 0.f;
            for (int j = t; j < cg; j += 256) {
                const int c = g * cg + j, plane = b * a.C + c;
                const float* pm = a.wmax + 2 * ((size_t)a.wave0[l] + (size_t)plane * cpp);
                float pg = 0.f, px = 0.f;
                for (int k = 0; k < cpp; ++k) { pg = fmaxf(pg, pm[2 * k]); px = fmaxf(px, pm[2 * k + 1]); }
                bnd = fmaxf(bnd, fabsf(r * (a.gamma ? a.gamma[c] : 1.f)) * pg + fabsf(r * m1) + fabsf(r * m2) * px);
            }
            bnd = wave_max(bnd) * 1.00001f;
            if (lane == 0) atomic_max_bits(a.bound_out, __builtin_bit_cast(unsigned, bnd));
        }
    }
}

// MODE 0: y = relu?(gamma*xhat + beta);  MODE 1: dx = rstd * (gamma*g - m1 - xhat*m2)
template <int MODE>
__global__ __launch_bounds__(256) void gg_apply_kernel(GgArgs a) {
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= a.nwaves) return;
    const int lane = threadIdx.x & 63;
    const GgWhere q = gg_locate(a, w);
    const int HW = a.HW[q.l];
    const float* __restrict__ px = a.x[q.l] + (size_t)q.plane * HW;
    const float* __restrict__ pd = MODE == 1 ? a.dy[q.l] + (size_t)q.plane * HW : nullptr;
    float* __restrict__ po = a.out[q.l] + (size_t)q.plane * HW;
    const int b = q.plane / a.C, c = q.plane % a.C, cg = a.C / a.G;
    const int seg = (q.l * a.B + b) * a.G + c / cg;
    const float mu = a.stats[2 * seg], r = a.stats[2 * seg + 1];
    const float ga = a.gamma ? a.gamma[c] : 1.f, be = a.beta ? a.beta[c] : 0.f;
    float m1 = 0.f, m2 = 0.f;
    if (MODE == 1) { m1 = a.bstats[2 * seg]; m2 = a.bstats[2 * seg + 1]; }
    auto f = [&](float x, float d) -> float {
        const float xh = gg_hat(x, mu, r);
        const float y = gg_affine(xh, ga, be);
        if (MODE == 0) return a.relu ? fmaxf(y, 0.f) : y;
        const float g = (a.relu && !(y > 0.f)) ? 0.f : d;
        return r * (ga * g - m1 - xh * m2);
    };
    const int e0 = q.chunk * a.chunk[q.l], e1 = min(HW, e0 + a.chunk[q.l]);
    if ((HW & 3) == 0) {
        for (int e = e0 + lane * 4; e < e1; e += 1024) {
            float4 vx[4], vd[4];
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * 256;
                vd[u] = make_float4(0, 0, 0, 0);
                if (ee < e1) { vx[u] = ldg_stream4(px + ee); if (MODE == 1) vd[u] = ldg_stream4(pd + ee); }
            }
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * 256;
                if (ee >= e1) continue;
                *reinterpret_cast<float4*>(po + ee) = make_float4(f(vx[u].x, vd[u].x), f(vx[u].y, vd[u].y), f(vx[u].z, vd[u].z),
                                                                  f(vx[u].w, vd[u].w));
            }
        }
    } else {
        for (int e = e0 + lane; e < e1; e += 64) po[e] = f(px[e], MODE == 1 ? pd[e] : 0.f);
    }
}

static int gg_fill(GgArgs& a, const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C, int G, int relu) {
    if (!x_host || !level_hw_host || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 1 || G < 1 || C % G) return LGD_EINVAL;
    a.L = L; a.B = B; a.C = C; a.G = G; a.relu = relu ? 1 : 0;
    a.wmax = nullptr; a.bound_out = nullptr;
    long long w = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.x[l] = nullptr; a.dy[l] = nullptr; a.out[l] = nullptr;
        a.wave0[l] = (int)w;
        if (l < L) {
            if (!x_host[l] || level_hw_host[2 * l] < 1 || level_hw_host[2 * l + 1] < 1) return LGD_EINVAL;
            a.x[l] = x_host[l];
            a.HW[l] = level_hw_host[2 * l] * level_hw_host[2 * l + 1];
            a.cpp[l] = (a.HW[l] + kGgChunk - 1) / kGgChunk;
            a.chunk[l] = ((a.HW[l] + a.cpp[l] - 1) / a.cpp[l] + 3) & ~3;
            w += (long long)B * C * a.cpp[l];
        } else { a.HW[l] = 0; a.cpp[l] = 1; a.chunk[l] = 4; }
    }
    if (w > 0x7fffffffLL) return LGD_EINVAL;
    a.wave0[LGD_MAX_LEVELS] = (int)w;
    a.nwaves = (int)w;
    a.gamma = a.beta = nullptr; a.ws = nullptr; a.stats = a.bstats = a.plane_sums = nullptr; a.affine = nullptr; a.coef = nullptr;
    return LGD_OK;
}

}  // namespace lgd

extern "C" {

size_t lgd_gn_group_ws_doubles(const int32_t* level_hw_host, int L, int B, int C) {
    size_t w = 0;
    for (int l = 0; l < L; ++l) {
        const int hw = level_hw_host[2 * l] * level_hw_host[2 * l + 1];
        w += (size_t)B * C * ((hw + lgd::kGgChunk - 1) / lgd::kGgChunk);
    }
    return 2 * w;
}

int lgd_gn_group_fwd(const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C, int G, const float* gamma,
                     const float* beta, int relu, double* ws, float* stats, float* const* y_host, void* stream) {
    lgd::GgArgs a;
    if (lgd::gg_fill(a, x_host, level_hw_host, L, B, C, G, relu) != LGD_OK || !ws || !stats || !y_host) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!y_host[l]) return LGD_EINVAL; a.out[l] = y_host[l]; }
    a.gamma = gamma; a.beta = beta; a.ws = ws; a.stats = stats;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((a.nwaves + 3) / 4);
    LGD_LAUNCH("gn_group_stats_kernel", lgd::gg_stats_kernel<0>, grid, dim3(256), 0, s, a);
    LGD_LAUNCH("gn_group_finalize_kernel", lgd::gg_finalize_kernel<0>, dim3(L * B * G), dim3(256), 0, s, a);
    LGD_LAUNCH("gn_group_apply_kernel", lgd::gg_apply_kernel<0>, grid, dim3(256), 0, s, a);
    return lgd::check_launch();
}

int lgd_gn_group_stats_affine(const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C, int G, const float* gamma,
                              const float* beta, double* ws, float* stats, float* affine, void* stream) {
    lgd::GgArgs a;
    if (lgd::gg_fill(a, x_host, level_hw_host, L, B, C, G, 1) != LGD_OK || !ws || !stats || !affine) return LGD_EINVAL;
    a.gamma = gamma; a.beta = beta; a.ws = ws; a.stats = stats; a.affine = affine;
    hipStream_t s = (hipStream_t)stream;
    LGD_LAUNCH("gn_group_stats_kernel", lgd::gg_stats_kernel<0>, dim3((a.nwaves + 3) / 4), dim3(256), 0, s, a);
    LGD_LAUNCH("gn_group_finalize_kernel", lgd::gg_finalize_kernel<0>, dim3(L * B * G), dim3(256), 0, s, a);
    return lgd::check_launch();
}

int lgd_gn_group_bwd(const float* const* x_host, const float* const* dy_host, const int32_t* level_hw_host, int L, int B, int C,
                     int G, const float* gamma, const float* beta, int relu, const float* stats, double* ws, float* bstats,
                     float* plane_sums, float* const* dx_host, void* stream) {
    lgd::GgArgs a;
    if (lgd::gg_fill(a, x_host, level_hw_host, L, B, C, G, relu) != LGD_OK || !dy_host || !stats || !ws || !bstats ||
        !plane_sums || !dx_host)
        return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!dy_host[l] || !dx_host[l]) return LGD_EINVAL;
        a.dy[l] = dy_host[l]; a.out[l] = dx_host[l];
    }
    a.gamma = gamma; a.beta = beta; a.ws = ws; a.stats = const_cast<float*>(stats); a.bstats = bstats; a.plane_sums = plane_sums;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((a.nwaves + 3) / 4);
    LGD_LAUNCH("gn_group_bwd_stats_kernel", lgd::gg_stats_kernel<1>, grid, dim3(256), 0, s, a);
    LGD_LAUNCH("gn_group_bwd_finalize_kernel", lgd::gg_finalize_kernel<1>, dim3(L * B * G), dim3(256), 0, s, a);
    LGD_LAUNCH("gn_group_bwd_apply_kernel", lgd::gg_apply_kernel<1>, grid, dim3(256), 0, s, a);
    return lgd::check_launch();
}

int lgd_gn_group_bwd_coef(const float* const* x_host, const float* const* dy_host, const int32_t* level_hw_host, int L, int B, int C,
                          int G, const float* gamma, const float* stats, double* ws, float* bstats, float* plane_sums, float* coef,
                          float* wmax, uint32_t* bound_out, void* stream) {
    lgd::GgArgs a;
    if (lgd::gg_fill(a, x_host, level_hw_host, L, B, C, G, 0) != LGD_OK || !dy_host || !stats || !ws || !bstats || !plane_sums || !coef)
        return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!dy_host[l]) return LGD_EINVAL;
        a.dy[l] = dy_host[l];
    }
    a.gamma = gamma; a.ws = ws; a.stats = const_cast<float*>(stats); a.bstats = bstats; a.plane_sums = plane_sums; a.coef = coef;
    a.wmax = wmax; a.bound_out = (wmax && bound_out) ? bound_out : nullptr;
    hipStream_t s = (hipStream_t)stream;
    LGD_LAUNCH("gn_group_bwd_stats_kernel", lgd::gg_stats_kernel<1>, dim3((a.nwaves + 3) / 4), dim3(256), 0, s, a);
    LGD_LAUNCH("gn_group_bwd_finalize_kernel", lgd::gg_finalize_kernel<1>, dim3(L * B * G), dim3(256), 0, s, a);
    return lgd::check_launch();
}

}  // extern "C"
