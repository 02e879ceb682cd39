// K1 / K3: box-sum (mask pooling fwd, render bwd) and box-paint (render fwd, mask pooling bwd).
//
// Reference arithmetic being replaced (dense fp32 GEMMs against materialised 0/1 masks):
//   [ref: dynamic_teacher.py:95-100]   pool = mask_b (Ni,HW) @ feat_b(C,HW)^T ; / max(mask.sum(-1),1)
//   [ref: dynamic_teacher.py:137,173]  warp = proj^T (C,Ni) @ mask_b (Ni,HW)
//
// MI355X design (HBM-bound: one pass over a pyramid, P = B*C*sum(HW)*4 bytes):
//   * one wave64 per (level, image, channel) plane; lane l owns VW adjacent columns, so a row is
//     one coalesced 16-byte-per-lane load (VW=4 when W%4==0); the row index is wave-uniform;
//   * the masks are axis-aligned rectangles, so rows are cut into BANDS inside which the set of
//     covering boxes is constant (box_geom.hip).  box_sum adds the rows of a band column-wise (one
//     VALU add per element) and only then applies each active box's column interval; box_paint
//     composes one row pattern per band and streams it to every row of the band;
//   * lane n of the wave holds box n's rectangle/accumulator (64 boxes per pass), band activity is a
//     single v_cmp ballot, box parameters travel by v_readlane, sums by DPP -- no LDS, no atomics,
//     fixed reduction order (bit-reproducible run to run).
#include <cstdlib>
#include "common.h"

namespace lgd {

struct BoxArgs {
    const float* in[LGD_MAX_LEVELS];   // box_sum: feature maps
    float* out[LGD_MAX_LEVELS];        // box_paint: painted maps
    int H[LGD_MAX_LEVELS], W[LGD_MAX_LEVELS];
    int blk0[LGD_MAX_LEVELS + 1];      // first block of each dispatch slot
    int lev[LGD_MAX_LEVELS];           // level handled by slot i (largest planes first)
    const float* vals;                 // box_paint: [L][T][C]
    float* pooled;                     // box_sum:   [L][T][C]
    const int32_t* img_off;
    const int32_t* geom;
    int L, B, C, T, max_n, normalize, skip_last;
    // fused GroupNorm(1)+ReLU on the fly (gn_pool): y = relu((x - mean_b) * rstd_b) is what gets pooled
    const float* gn_stats;             // [L*B][2] mean, rstd (nullptr: plain box_sum)
    const float* gn_bstats;            // [L*B][2] m1, m2 (backward apply)
    const float* gx[LGD_MAX_LEVELS];   // backward: conv output x
    double* ws;                        // backward stats: [L][B*C][2] per-plane partial sums
};

struct Plane { int l, b, c, H, W, t0, n, nbp; const int32_t* rects; const int32_t* bands; };

// ppb = planes per workgroup: 4 (one wave per plane, box_paint) or 1 (four waves share a plane, box_sum)
__device__ __forceinline__ Plane locate(const BoxArgs& a, int ppb) {
    Plane p;
    int slot = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) slot += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    const int l = a.lev[slot];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the plane bookkeeping on the scalar unit
    // ppb planes per workgroup: each wave owns ppb/4 consecutive channel planes (ppb == 1: all four waves share one)
    const int plane = ppb >= 4 ? ((int)blockIdx.x - a.blk0[slot]) * ppb + wave * (ppb / 4) : (int)blockIdx.x - a.blk0[slot];
    p.l = l; p.b = plane / a.C; p.c = plane % a.C;
    p.H = a.H[l]; p.W = a.W[l];
    p.t0 = __builtin_amdgcn_readfirstlane(a.img_off[p.b]);
    p.n = __builtin_amdgcn_readfirstlane(a.img_off[p.b + 1]) - p.t0;
    p.rects = a.geom + geom_rects_off() + ((size_t)l * a.B + p.b) * a.max_n * 4;
    p.nbp = __builtin_amdgcn_readfirstlane(a.geom[geom_nbp_off(a.L, a.B, a.max_n) + (size_t)l * a.B + p.b]);
    p.bands = a.geom + geom_bands_off(a.L, a.B, a.max_n) + ((size_t)l * a.B + p.b) * geom_maxbp(a.max_n);
    return p;
}

struct LaneBox { int x0, x1, y0, y1; };  // lane-resident rectangle of box (pass*64 + lane); empty: x1 < x0

__device__ __forceinline__ LaneBox load_lane_box_at(const Plane& p, int n, int skip_last) {
    LaneBox r{0, -1, 0, -1};
    if (n < p.n && !(skip_last && n == p.n - 1)) {
        const int4 q = reinterpret_cast<const int4*>(p.rects)[n];
        r.x0 = q.x; r.x1 = q.y; r.y0 = q.z; r.y1 = q.w;
    }
    return r;
}
__device__ __forceinline__ LaneBox load_lane_box(const Plane& p, int pass, int lane, int skip_last) {
    const int n = pass * 64 + lane;
    LaneBox r{0, -1, 0, -1};
    if (n < p.n && !(skip_last && n == p.n - 1)) {
        const int4 q = reinterpret_cast<const int4*>(p.rects)[n];
        r.x0 = q.x; r.x1 = q.y; r.y0 = q.z; r.y1 = q.w;
    }
    return r;
}

// ------------------------------------------------------------------------------------------- box_sum
// One wave per plane.  Rows stream through a two-deep register pipeline of FIXED 4-row groups that ignores band
// boundaries: every group is exactly 4 loads, so the compiler can keep one group in flight with a counted
// s_waitcnt while the other is reduced (62 VGPRs -> 8 waves/SIMD: the whole grid is co-resident).  (The first
// version loaded band by band and exposed one HBM latency per band: ~20 x 2.5 us per p3 wave, 3.3 TB/s.
// Variable-length groups force s_waitcnt vmcnt(0) and serialise the pipeline again: 1.9 TB/s measured.)
// Band bookkeeping happens at consume time with wave-uniform control flow; the band table and the rectangles
// live in registers (lane k <- bands[k], lane n <- box n; v_readlane), so the loop touches memory only for rows.
// Band flush: every active box adds one masked partial per lane into that lane's private LDS slot
// sacc[box][lane] (conflict-free); the 64 -> 1 reductions happen once per (plane, box) at the end.
// Measured alternatives that were SLOWER on MI355X (kept out): 4 waves per plane by row interleave or row quarters
// (80-110 us: per-wave fixed costs x4), small levels dispatched first (56 us), an LDS-tiled variant with full-width
// 1 KB loads and per-box rectangle sums out of LDS (101 us: 4x the load instructions on the small levels, same
// VALU/SALU count).  SQ counters of this version: 18.5 M VALU + 14.3 M SALU instructions for 0.5 M loads per launch,
// 28 % of wave cycles issuing, 26 % waiting on memory -- it is issue/latency-bound, not HBM-bound.
// NP = channel planes per wave: planes c, c+1 of one image share every piece of bookkeeping (band walk, activity
// ballots, box column tests, v_readlane traffic), which is what bounds this kernel -- not bytes.
template <int VW, int NP>
__device__ __forceinline__ void box_sum_plane(const BoxArgs& a, const Plane& p, float* sacc /* [NP][nb][64] of this wave */, int nb) {
    constexpr int G = 4;
    const int lane = threadIdx.x & 63;
    const size_t psz = (size_t)p.H * p.W;
    const float* __restrict__ src = a.in[p.l] + ((size_t)p.b * a.C + p.c) * psz;
    const int npass = (p.n + nb - 1) / nb;   // nb = boxes per pass (LDS budget), <= 64
    const int bandreg = lane < p.nbp ? p.bands[lane] : p.H;  // nbp <= 64 is the fast path
    auto band = [&](int k) {  // wave-uniform by construction: say so, or every band test becomes an exec-masked vector loop
        return __builtin_amdgcn_readfirstlane(p.nbp <= 64 ? __builtin_amdgcn_readlane(bandreg, k) : p.bands[k]);
    };
    const bool gn = a.gn_stats != nullptr;
    const float gmu = gn ? a.gn_stats[2 * (p.l * a.B + p.b)] : 0.f, grs = gn ? a.gn_stats[2 * (p.l * a.B + p.b) + 1] : 1.f;
    for (int pass = 0; pass < npass; ++pass) {
        LaneBox bx{0, -1, 0, -1};
        if (lane < nb) bx = load_lane_box_at(p, pass * nb + lane, a.skip_last);
        for (int n = 0; n < NP * nb; ++n) sacc[n * 64 + lane] = 0.f;
        // rows below ylo / above yhi are covered by no box of this pass: never fetched
        int ylo = p.H, yhi = -1;
        {
            unsigned long long live = __ballot(bx.x1 >= bx.x0);
            while (live) {
                const int n = __builtin_ctzll(live);
                live &= live - 1;
                ylo = min(ylo, __builtin_amdgcn_readlane(bx.y0, n));
                yhi = max(yhi, __builtin_amdgcn_readlane(bx.y1, n));
            }
        }
        for (int xc = 0; yhi >= ylo && xc < p.W; xc += 64 * VW) {
            const int xl = xc + lane * VW;
            const bool on = xl < p.W;  // VW | W, so a lane's vector is wholly inside or outside the row
            const float* col = src + (on ? xl : 0);
            auto band_act = [&](int ya) {
                return __ballot(bx.x1 >= bx.x0 && bx.y0 <= ya && ya <= bx.y1 && bx.x1 >= xc && bx.x0 < xc + 64 * VW);
            };
            int k = 0;
            while (band(k + 1) <= ylo) ++k;      // band containing ylo
            int yb = band(k + 1);
            unsigned long long act = band_act(band(k));
            float cs[NP][VW];
            #pragma unroll
            for (int q = 0; q < NP; ++q)
                #pragma unroll
                for (int j = 0; j < VW; ++j) cs[q][j] = 0.f;
            auto flush = [&]() {
                unsigned long long m = act;
                while (m) {
                    const int n = __builtin_ctzll(m);
                    m &= m - 1;
                    const int bx0 = __builtin_amdgcn_readlane(bx.x0, n), bx1 = __builtin_amdgcn_readlane(bx.x1, n);
                    float part[NP];
                    #pragma unroll
                    for (int q = 0; q < NP; ++q) part[q] = 0.f;
                    #pragma unroll
                    for (int j = 0; j < VW; ++j) {
                        const bool in = xl + j >= bx0 && xl + j <= bx1;
                        #pragma unroll
                        for (int q = 0; q < NP; ++q) part[q] += in ? cs[q][j] : 0.f;
                    }
                    #pragma unroll
                    for (int q = 0; q < NP; ++q) sacc[(q * nb + n) * 64 + lane] += part[q];
                }
                #pragma unroll
                for (int q = 0; q < NP; ++q)
                    #pragma unroll
                    for (int j = 0; j < VW; ++j) cs[q][j] = 0.f;
            };
            auto issue = [&](Vec<VW> (*v)[G], int y0) {  // always NP*G loads (rows clamped into the plane)
                #pragma unroll
                for (int u = 0; u < G; ++u) {
                    const float* rp = col + (size_t)min(y0 + u, p.H - 1) * p.W;
                    #pragma unroll
                    for (int q = 0; q < NP; ++q) v[q][u] = vload<VW>(rp + q * psz);
                }
            };
            auto consume = [&](Vec<VW> (*v)[G], int y0) {
                #pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int y = y0 + u;
                    if (y <= yhi) {
                        while (y == yb) {  // wave-uniform: close the band, open the next
                            flush();
                            ++k;
                            yb = band(k + 1);
                            act = band_act(y);
                        }
                        if (act && on) {
                            #pragma unroll
                            for (int q = 0; q < NP; ++q)
                                #pragma unroll
                                for (int j = 0; j < VW; ++j)
                                    cs[q][j] += gn ? fmaxf(__fmul_rn(__fsub_rn(v[q][u].v[j], gmu), grs), 0.f) : v[q][u].v[j];
                        }
                    }
                }
            };
            Vec<VW> va[NP][G], vb[NP][G];
            issue(va, ylo);
            for (int y0 = ylo; y0 <= yhi; y0 += 2 * G) {
                issue(vb, y0 + G);
                consume(va, y0);
                issue(va, y0 + 2 * G);
                consume(vb, y0 + G);
            }
            flush();
        }
        // one 64 -> 1 reduction per (plane, box) of the pass; lane n keeps box n's totals
        float mine[NP];
        #pragma unroll
        for (int q = 0; q < NP; ++q) mine[q] = 0.f;
        const int nlive = min(nb, p.n - pass * nb);
        for (int n = 0; n < nlive; ++n) {
            #pragma unroll
            for (int q = 0; q < NP; ++q) {
                const float tot = wave_sum(sacc[(q * nb + n) * 64 + lane]);
                if (lane == n) mine[q] = tot;
            }
        }
        if (lane < nlive) {
            const float cnt = (bx.x1 >= bx.x0) ? (float)((bx.x1 - bx.x0 + 1) * (bx.y1 - bx.y0 + 1)) : 0.f;
            #pragma unroll
            for (int q = 0; q < NP; ++q) {
                float o = mine[q];
                if (a.normalize) o = o / fmaxf(cnt, 1.f);  // [ref: dynamic_teacher.py:97-100]
                a.pooled[((size_t)p.l * a.T + p.t0 + pass * nb + lane) * a.C + p.c + q] = o;
            }
        }
    }
}

template <int NP>  // channel planes per wave
__global__ __launch_bounds__(256) void box_sum_kernel(BoxArgs a, int nb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [4 waves][NP][nb][64]
    const Plane p = locate(a, 4 * NP);
    float* sacc = smem + (size_t)(threadIdx.x >> 6) * NP * nb * 64;
    if ((p.W & 3) == 0) box_sum_plane<4, NP>(a, p, sacc, nb);
    else if ((p.W & 1) == 0) box_sum_plane<2, NP>(a, p, sacc, nb);
    else box_sum_plane<1, NP>(a, p, sacc, nb);
}

// planes per wave: as many as the channel count allows (the shared bookkeeping is the cost centre)
static int sum_planes(int C) {
    const char* e = getenv("LGD_SUM_NP");
    const int want = e ? atoi(e) : 2;
    if (want >= 4 && C % 16 == 0) return 4;
    if (want >= 2 && C % 8 == 0) return 2;
    return 1;
}
static void launch_box_sum(const char* name, const BoxArgs& a, int np, int nblk, int nb, hipStream_t s) {
    const size_t smem = (size_t)4 * np * nb * 64 * sizeof(float);
    if (np == 4) LGD_LAUNCH(name, box_sum_kernel<4>, dim3(nblk), dim3(256), smem, s, a, nb);
    else if (np == 2) LGD_LAUNCH(name, box_sum_kernel<2>, dim3(nblk), dim3(256), smem, s, a, nb);
    else LGD_LAUNCH(name, box_sum_kernel<1>, dim3(nblk), dim3(256), smem, s, a, nb);
}

// ------------------------------------------------------------------------------------------- box_paint
template <int VW>
__device__ __forceinline__ void box_paint_plane(const BoxArgs& a, const Plane& p) {
    const int lane = threadIdx.x & 63;
    float* __restrict__ dst = a.out[p.l] + ((size_t)p.b * a.C + p.c) * p.H * p.W;
    const float* __restrict__ vals = a.vals + ((size_t)p.l * a.T + p.t0) * a.C + p.c;
    const int npass = (p.n + 63) >> 6;

    auto lane_val = [&](const LaneBox& bx, int pass) -> float {
        const int n = pass * 64 + lane;
        float v = 0.f;
        if (bx.x1 >= bx.x0) {  // only live boxes are read (the skipped context row may hold anything)
            v = vals[(size_t)n * a.C];
            if (a.normalize) v = v / fmaxf((float)((bx.x1 - bx.x0 + 1) * (bx.y1 - bx.y0 + 1)), 1.f);
        }
        return v;
    };
    // common case (<= 64 boxes per image): rectangles and values stay in registers for the whole plane
    LaneBox bx0 = load_lane_box(p, 0, lane, a.skip_last);
    float val0 = lane_val(bx0, 0);

    for (int xc = 0; xc < p.W; xc += 64 * VW) {
        const int xl = xc + lane * VW;
        const bool on = xl < p.W;
        float* col = dst + xl;
        for (int k = 0; k + 1 < p.nbp; ++k) {
            const int ya = __builtin_amdgcn_readfirstlane(p.bands[k]), yb = __builtin_amdgcn_readfirstlane(p.bands[k + 1]);
            Vec<VW> pv;
            #pragma unroll
            for (int j = 0; j < VW; ++j) pv.v[j] = 0.f;
            for (int pass = 0; pass < npass; ++pass) {
                LaneBox bx = bx0;
                float val = val0;
                if (pass > 0) { bx = load_lane_box(p, pass, lane, a.skip_last); val = lane_val(bx, pass); }
                unsigned long long act = __ballot(bx.x1 >= bx.x0 && bx.y0 <= ya && ya <= bx.y1);
                while (act) {
                    const int n = __builtin_ctzll(act);
                    act &= act - 1;
                    const int q0 = __builtin_amdgcn_readlane(bx.x0, n), q1 = __builtin_amdgcn_readlane(bx.x1, n);
                    const float v = readlane_f32(val, n);
                    #pragma unroll
                    for (int j = 0; j < VW; ++j) pv.v[j] += (xl + j >= q0 && xl + j <= q1) ? v : 0.f;
                }
            }
            if (on) {
                for (int y = ya; y < yb; ++y) vstore<VW>(col + (size_t)y * p.W, pv);
            }
        }
    }
}

__global__ __launch_bounds__(256) void box_paint_kernel(BoxArgs a) {
    const Plane p = locate(a, 4);
    if ((p.W & 3) == 0) box_paint_plane<4>(a, p);
    else if ((p.W & 1) == 0) box_paint_plane<2>(a, p);
    else box_paint_plane<1>(a, p);
}

// ------------------------------------------------------------------------------------------- gn_pool backward
// d/dx of  pool(relu(GN1(x))):  dy = paint(dpool / count) restricted to y > 0, then the GroupNorm(1) backward
// dx = rstd * (g - m1 - xhat * m2), m1 = mean(g), m2 = mean(g * xhat) over the sample.  dy is never materialised:
// every band's row pattern pv is composed from the active boxes (as in box_paint) and applied while x streams by.
//   MODE 0: per-plane partial sums of g and g*xhat (fp64) -> ws        (reads x once)
//   MODE 1: dx                                                         (reads x once, writes dx)
template <int VW, int MODE>
__device__ __forceinline__ void gn_pool_bwd_plane(const BoxArgs& a, const Plane& p) {
    const int lane = threadIdx.x & 63;
    const size_t base = ((size_t)p.b * a.C + p.c) * p.H * p.W;
    const float* __restrict__ src = a.gx[p.l] + base;
    float* __restrict__ dst = MODE == 1 ? a.out[p.l] + base : nullptr;
    const float* __restrict__ vals = a.vals + ((size_t)p.l * a.T + p.t0) * a.C + p.c;
    const int seg = p.l * a.B + p.b;
    const float mu = a.gn_stats[2 * seg], rs = a.gn_stats[2 * seg + 1];
    float m1 = 0.f, m2 = 0.f;
    if (MODE == 1) { m1 = a.gn_bstats[2 * seg]; m2 = a.gn_bstats[2 * seg + 1]; }
    const int npass = (p.n + 63) >> 6;
    auto lane_val = [&](const LaneBox& bx, int pass) -> float {
        float v = 0.f;
        if (bx.x1 >= bx.x0) v = vals[(size_t)(pass * 64 + lane) * a.C] / fmaxf((float)((bx.x1 - bx.x0 + 1) * (bx.y1 - bx.y0 + 1)), 1.f);
        return v;
    };
    const LaneBox bx0 = load_lane_box(p, 0, lane, 0);
    const float val0 = lane_val(bx0, 0);
    double s1 = 0.0, s2 = 0.0;
    for (int xc = 0; xc < p.W; xc += 64 * VW) {
        const int xl = xc + lane * VW;
        const bool on = xl < p.W;
        const float* col = src + (on ? xl : 0);
        for (int k = 0; k + 1 < p.nbp; ++k) {
            const int ya = __builtin_amdgcn_readfirstlane(p.bands[k]), yb = __builtin_amdgcn_readfirstlane(p.bands[k + 1]);
            float pv[VW];
            #pragma unroll
            for (int j = 0; j < VW; ++j) pv[j] = 0.f;
            bool any = false;
            for (int pass = 0; pass < npass; ++pass) {
                LaneBox bx = bx0;
                float val = val0;
                if (pass > 0) { bx = load_lane_box(p, pass, lane, 0); val = lane_val(bx, pass); }
                unsigned long long act = __ballot(bx.x1 >= bx.x0 && bx.y0 <= ya && ya <= bx.y1);
                any |= act != 0ull;
                while (act) {
                    const int n = __builtin_ctzll(act);
                    act &= act - 1;
                    const int q0 = __builtin_amdgcn_readlane(bx.x0, n), q1 = __builtin_amdgcn_readlane(bx.x1, n);
                    const float v = readlane_f32(val, n);
                    #pragma unroll
                    for (int j = 0; j < VW; ++j) pv[j] += (xl + j >= q0 && xl + j <= q1) ? v : 0.f;
                }
            }
            if (MODE == 0 && !any) continue;  // uncovered rows contribute g = 0 to both sums: not even read
            if (!on) continue;
            for (int y0 = ya; y0 < yb; y0 += 4) {
                Vec<VW> v[4];
                #pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = vload<VW>(col + (size_t)min(y0 + u, yb - 1) * p.W);
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (y0 + u >= yb) continue;
                    Vec<VW> o;
                    float t1 = 0.f, t2 = 0.f;
                    #pragma unroll
                    for (int j = 0; j < VW; ++j) {
                        const float xh = __fmul_rn(__fsub_rn(v[u].v[j], mu), rs);
                        const float g = xh > 0.f ? pv[j] : 0.f;
                        if (MODE == 0) { t1 += g; t2 = fmaf(g, xh, t2); }
                        else o.v[j] = rs * (g - m1 - xh * m2);
                    }
                    if (MODE == 0) { s1 += (double)t1; s2 += (double)t2; }
                    else vstore<VW>(dst + xl + (size_t)(y0 + u) * p.W, o);
                }
            }
        }
    }
    if (MODE == 0) {
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) {
            double* o = a.ws + 2 * (((size_t)p.l * a.B + p.b) * a.C + p.c);
            o[0] = s1; o[1] = s2;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void gn_pool_bwd_kernel(BoxArgs a) {
    const Plane p = locate(a, 4);
    if ((p.W & 3) == 0) gn_pool_bwd_plane<4, MODE>(a, p);
    else if ((p.W & 1) == 0) gn_pool_bwd_plane<2, MODE>(a, p);
    else gn_pool_bwd_plane<1, MODE>(a, p);
}

// per (level, image): fold the C per-plane partials -> m1 = mean(g), m2 = mean(g*xhat)
__global__ __launch_bounds__(256) void gn_pool_bwd_finalize_kernel(BoxArgs a, float* bstats) {
    __shared__ double red[8];
    const int seg = blockIdx.x, l = seg / a.B;
    const double* p = a.ws + 2 * (size_t)seg * a.C;
    double s1 = 0, s2 = 0;
    for (int c = threadIdx.x; c < a.C; c += 256) { s1 += p[2 * c]; s2 += p[2 * c + 1]; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = s1; red[2 * (threadIdx.x >> 6) + 1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = (double)a.C * a.H[l] * a.W[l];
        bstats[2 * seg] = (float)(((red[0] + red[2]) + (red[4] + red[6])) / n);
        bstats[2 * seg + 1] = (float)(((red[1] + red[3]) + (red[5] + red[7])) / n);
    }
}

static int fill_args(BoxArgs& a, const int32_t* level_hw_host, int L, int B, int C, int T, int max_n,
                     const int32_t* img_off, const int32_t* geom, int normalize, int skip_last, int ppb) {
    if (!level_hw_host || !img_off || !geom || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 4 || (C & 3) || T < 0)
        return LGD_EINVAL;
    a.L = L; a.B = B; a.C = C; a.T = T; a.max_n = max_n; a.normalize = normalize; a.skip_last = skip_last;
    a.img_off = img_off; a.geom = geom; a.vals = nullptr; a.pooled = nullptr;
    a.gn_stats = nullptr; a.gn_bstats = nullptr; a.ws = nullptr;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) a.gx[l] = nullptr;
    int blk = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.in[l] = nullptr; a.out[l] = nullptr;
        a.H[l] = l < L ? level_hw_host[2 * l] : 0;
        a.W[l] = l < L ? level_hw_host[2 * l + 1] : 0;
        a.lev[l] = 0;
    }
    // dispatch slots by DESCENDING plane size (stable): the long p3 waves must start first (small-first measured 15 % slower)
    int order[LGD_MAX_LEVELS];
    for (int i = 0; i < L; ++i) order[i] = i;
    for (int i = 1; i < L; ++i)
        for (int j = i; j > 0 && a.H[order[j]] * a.W[order[j]] > a.H[order[j - 1]] * a.W[order[j - 1]]; --j) {
            const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
        }
    for (int i = 0; i < LGD_MAX_LEVELS; ++i) {
        a.blk0[i] = blk;
        if (i < L) { a.lev[i] = order[i]; blk += B * C / ppb; }
    }
    a.blk0[LGD_MAX_LEVELS] = blk;
    return blk;
}

}  // namespace lgd

extern "C" {

int lgd_box_sum(const float* const* feats_host, const int32_t* level_hw_host, int L, int B, int C, int T, int max_n,
                const int32_t* img_off, const int32_t* geom, float* out, int normalize, int skip_last, void* stream) {
    lgd::BoxArgs a;
    const int np = lgd::sum_planes(C);
    const int nblk = lgd::fill_args(a, level_hw_host, L, B, C, T, max_n, img_off, geom, normalize, skip_last, 4 * np);
    if (nblk < 0 || !feats_host || !out) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!feats_host[l]) return LGD_EINVAL; a.in[l] = feats_host[l]; }
    a.pooled = out;
    if (T == 0) return LGD_OK;
    const int nb = max_n < 1 ? 1 : (max_n > 64 ? 64 : max_n);  // boxes per pass: 256 B of LDS per wave and box
    lgd::launch_box_sum("box_sum_kernel", a, np, nblk, nb, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_gn_pool_fwd(const float* const* x_host, const float* gn_stats, const int32_t* level_hw_host, int L, int B, int C, int T,
                    int max_n, const int32_t* img_off, const int32_t* geom, float* out, void* stream) {
    lgd::BoxArgs a;
    const int np = lgd::sum_planes(C);
    const int nblk = lgd::fill_args(a, level_hw_host, L, B, C, T, max_n, img_off, geom, 1, 0, 4 * np);
    if (nblk < 0 || !x_host || !gn_stats || !out) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!x_host[l]) return LGD_EINVAL; a.in[l] = x_host[l]; }
    a.pooled = out; a.gn_stats = gn_stats;
    if (T == 0) return LGD_OK;
    const int nb = max_n < 1 ? 1 : (max_n > 64 ? 64 : max_n);
    lgd::launch_box_sum("gn_pool_kernel", a, np, nblk, nb, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_gn_pool_bwd(const float* const* x_host, const float* gn_stats, const float* dpool, const int32_t* level_hw_host, int L,
                    int B, int C, int T, int max_n, const int32_t* img_off, const int32_t* geom, double* ws, float* bstats,
                    float* const* dx_host, void* stream) {
    lgd::BoxArgs a;
    const int nblk = lgd::fill_args(a, level_hw_host, L, B, C, T, max_n, img_off, geom, 1, 0, 4);
    if (nblk < 0 || !x_host || !gn_stats || !dpool || !ws || !bstats || !dx_host) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!x_host[l] || !dx_host[l]) return LGD_EINVAL;
        a.gx[l] = x_host[l]; a.out[l] = dx_host[l];
    }
    a.vals = dpool; a.gn_stats = gn_stats; a.gn_bstats = bstats; a.ws = ws;
    hipStream_t s = (hipStream_t)stream;
    LGD_LAUNCH("gn_pool_bwd_stats_kernel", lgd::gn_pool_bwd_kernel<0>, dim3(nblk), dim3(256), 0, s, a);
    LGD_LAUNCH("gn_pool_bwd_finalize_kernel", lgd::gn_pool_bwd_finalize_kernel, dim3(L * B), dim3(256), 0, s, a, bstats);
    LGD_LAUNCH("gn_pool_bwd_apply_kernel", lgd::gn_pool_bwd_kernel<1>, dim3(nblk), dim3(256), 0, s, a);
    return lgd::check_launch();
}

int lgd_box_paint(const float* vals, const int32_t* level_hw_host, int L, int B, int C, int T, int max_n,
                  const int32_t* img_off, const int32_t* geom, float* const* outs_host, int normalize, int skip_last,
                  void* stream) {
    lgd::BoxArgs a;
    const int nblk = lgd::fill_args(a, level_hw_host, L, B, C, T, max_n, img_off, geom, normalize, skip_last, 4);
    if (nblk < 0 || !outs_host || !vals) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!outs_host[l]) return LGD_EINVAL; a.out[l] = outs_host[l]; }
    a.vals = vals;
    LGD_LAUNCH("box_paint_kernel", lgd::box_paint_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
