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
//     single v_cmp ballot, box parameters travel by v_readlane; a band's column sums become box sums through one
//     wave-wide prefix sum (DPP) parked in wave-private LDS -- no atomics, fixed order (bit-reproducible run to run).
#include "common.h"

// rows per load group of the band-streaming kernels (two groups in flight per wave and plane)
#ifndef LGD_BAND_G
#define LGD_BAND_G 4
#endif

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

// ppb = planes per workgroup (a multiple of 4): each of the four waves owns ppb/4 consecutive channel planes
__device__ __forceinline__ Plane locate(const BoxArgs& a, int ppb) {
    Plane p;
    int slot = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) slot += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    const int l = a.lev[slot];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the plane bookkeeping on the scalar unit
    const int plane = ((int)blockIdx.x - a.blk0[slot]) * ppb + wave * (ppb / 4);
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
// One wave per NP channel planes (planes c, c+1 of one image share all band bookkeeping).  Rows stream through a two-deep register
// pipeline of FIXED 4-row groups that ignores band boundaries: every group is exactly NP*4 loads, so the compiler keeps one group in
// flight with a counted s_waitcnt while the other is reduced.  (Loading band by band exposed one HBM latency per band: ~20 x 2.5 us per
// p3 wave, 3.3 TB/s; variable-length groups force s_waitcnt vmcnt(0): 1.9 TB/s.)  Band bookkeeping happens at consume time with
// wave-uniform control flow; the band table and the rectangles live in registers (lane k <- bands[k], lane n <- box n).
// Band flush = ONE wave-wide prefix sum of the band's column sums per plane (fp64 across lanes, fp32 inside a lane's VW columns),
// parked in LDS; lane n then reads the prefix at its box's two column ends and adds the difference to box n's total in a register --
// all (<= 64) boxes of the image at once, cost independent of how many are active.  (Round 1's flush added one masked partial per
// ACTIVE box into LDS slots and reduced 64 -> 1 per box and plane at the end: same speed on box_sum, 8 % slower with the fused
// GroupNorm + ReLU, whose extra arithmetic competes for the same issue slots.)
// SQ counters (tools/sq_counters.sh): the kernel is ISSUE-bound, not HBM-bound -- 16 M VALU + 8 M SALU instructions per launch, the
// resident waves' issue shares add up to one SIMD; a row of p3 fills 42 of 64 lanes, the small levels 11-21.  Measured and rejected:
// splitting the rows of the big planes over the four waves of a workgroup (4x the waves, each with its own band walk and flushes):
// 42 -> 51-55 us; 4 waves per plane by row interleave, small levels first, an LDS-tiled variant (round 1): all slower.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64m(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_incl_scan(double v) {
    v += dpp_f64m<0x111, 0xf>(v);  // row_shr:1 .. 8: Kogge-Stone inside each row of 16 lanes (lanes without a source read 0)
    v += dpp_f64m<0x112, 0xf>(v);
    v += dpp_f64m<0x114, 0xf>(v);
    v += dpp_f64m<0x118, 0xf>(v);
    v += dpp_f64m<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_f64m<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_i32<0xB1>(v)); v = min(v, dpp_i32<0x4E>(v)); v = min(v, dpp_i32<0x141>(v)); v = min(v, dpp_i32<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, dpp_i32<0xB1>(v)); v = max(v, dpp_i32<0x4E>(v)); v = max(v, dpp_i32<0x141>(v)); v = max(v, dpp_i32<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

struct ScanScratch { double E[64]; float loc[256]; };   // per wave and plane: exclusive lane prefix, inclusive in-lane prefix by column

template <int VW, int NP>
__device__ __forceinline__ void box_sum_scan(const BoxArgs& a, const Plane& p, ScanScratch* sc) {
    constexpr int G = LGD_BAND_G;
    const int lane = threadIdx.x & 63;
    const size_t psz = (size_t)p.H * p.W;
    const float* __restrict__ src = a.in[p.l] + ((size_t)p.b * a.C + p.c) * psz;
    const int npass = (p.n + 63) >> 6;
    const int bandreg = lane < p.nbp ? p.bands[lane] : p.H;
    auto band = [&](int k) {  // wave-uniform by construction
        return __builtin_amdgcn_readfirstlane(p.nbp <= 64 ? __builtin_amdgcn_readlane(bandreg, k) : (k < p.nbp ? p.bands[k] : p.H));
    };
    const bool gn = a.gn_stats != nullptr;
    const float gmu = gn ? a.gn_stats[2 * (p.l * a.B + p.b)] : 0.f, grs = gn ? a.gn_stats[2 * (p.l * a.B + p.b) + 1] : 1.f;
    for (int pass = 0; pass < npass; ++pass) {
        const LaneBox bx = load_lane_box_at(p, pass * 64 + lane, a.skip_last);
        const bool live = bx.x1 >= bx.x0 && bx.y1 >= bx.y0;
        // rows any box of this pass covers: the others are never fetched
        const int ra = __builtin_amdgcn_readfirstlane(wave_min_i(live ? bx.y0 : p.H));
        const int rb = __builtin_amdgcn_readfirstlane(wave_max_i(live ? bx.y1 : -1) + 1);
        double acc[NP];
        #pragma unroll
        for (int q = 0; q < NP; ++q) acc[q] = 0.0;
        for (int xc = 0; rb > ra && xc < p.W; xc += 64 * VW) {
            const int xl = xc + lane * VW;
            const bool on = xl < p.W;  // VW | W, so a lane's vector is wholly inside or outside the row
            const float* col = src + (on ? xl : 0);
            const bool inchunk = live && bx.x1 >= xc && bx.x0 < xc + 64 * VW;
            int k = 0;
            while (band(k + 1) <= ra) ++k;       // band containing ra
            int yb = band(k + 1);
            bool mine = inchunk && bx.y0 <= ra && ra <= bx.y1;   // this lane's box is active in the current band
            bool any = __ballot(mine) != 0ull;
            float cs[NP][VW];
            #pragma unroll
            for (int q = 0; q < NP; ++q)
                #pragma unroll
                for (int j = 0; j < VW; ++j) cs[q][j] = 0.f;
            auto flush = [&]() {
                if (!any) return;   // wave-uniform; cs is still zero
                const int xa = max(bx.x0, xc) - xc, xb = min(bx.x1, xc + 64 * VW - 1) - xc;   // chunk-relative column ends
                #pragma unroll
                for (int q = 0; q < NP; ++q) {
                    float pre[VW];
                    pre[0] = cs[q][0];
                    #pragma unroll
                    for (int j = 1; j < VW; ++j) pre[j] = pre[j - 1] + cs[q][j];
                    const double tot = (double)pre[VW - 1];
                    sc[q].E[lane] = wave_incl_scan(tot) - tot;
                    #pragma unroll
                    for (int j = 0; j < VW; ++j) sc[q].loc[lane * VW + j] = pre[j];
                    #pragma unroll
                    for (int j = 0; j < VW; ++j) cs[q][j] = 0.f;
                }
                if (mine) {   // same wave: the LDS queue is in order, the writes above are visible
                    #pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const double hi = sc[q].E[xb / VW] + (double)sc[q].loc[xb];
                        const double lo = xa > 0 ? sc[q].E[(xa - 1) / VW] + (double)sc[q].loc[xa - 1] : 0.0;
                        acc[q] += hi - lo;
                    }
                }
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
                    if (y < rb) {
                        while (y == yb) {  // wave-uniform: close the band, open the next
                            flush();
                            ++k;
                            yb = band(k + 1);
                            mine = inchunk && bx.y0 <= y && y <= bx.y1;
                            any = __ballot(mine) != 0ull;
                        }
                        if (any && on) {
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
            issue(va, ra);
            for (int y0 = ra; y0 < rb; y0 += 2 * G) {
                issue(vb, y0 + G);
                consume(va, y0);
                issue(va, y0 + 2 * G);
                consume(vb, y0 + G);
            }
            flush();
        }
        if (pass * 64 + lane < p.n) {
            const float cnt = live ? (float)((bx.x1 - bx.x0 + 1) * (bx.y1 - bx.y0 + 1)) : 0.f;
            #pragma unroll
            for (int q = 0; q < NP; ++q) {
                float o = (float)acc[q];
                if (a.normalize) o = o / fmaxf(cnt, 1.f);  // [ref: dynamic_teacher.py:97-100]
                a.pooled[((size_t)p.l * a.T + p.t0 + pass * 64 + lane) * a.C + p.c + q] = o;
            }
        }
    }
}

template <int NP>  // channel planes per wave
__global__ __launch_bounds__(256) void box_sum_kernel(BoxArgs a) {
    __shared__ ScanScratch sc[4][NP];
    const Plane p = locate(a, 4 * NP);
    ScanScratch* mine = sc[threadIdx.x >> 6];
    if ((p.W & 3) == 0) box_sum_scan<4, NP>(a, p, mine);
    else if ((p.W & 1) == 0) box_sum_scan<2, NP>(a, p, mine);
    else box_sum_scan<1, NP>(a, p, mine);
}
static void launch_box_sum(const char* name, const BoxArgs& a, int np, int nblk, hipStream_t s) {
    if (np == 2) LGD_LAUNCH(name, box_sum_kernel<2>, dim3(nblk), dim3(256), 0, s, a);
    else LGD_LAUNCH(name, box_sum_kernel<1>, dim3(nblk), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------- box_paint / gn_pool backward
// One skeleton for the three kernels that apply a per-band ROW PATTERN pv[x] = sum of the values of the boxes covering (band, x):
//   MODE 2  box_paint          dst = pv                                     (render fwd, mask pooling bwd; writes only)
//   MODE 0  gn_pool bwd stats  per-plane sums of g and g*xhat, g = pv where relu(GN1(x)) > 0   (reads x)
//   MODE 1  gn_pool bwd apply  dx = rstd * (g - m1 - xhat * m2)                                (reads x, writes dx)
// d/dx of pool(relu(GN1(x))): dy = paint(dpool / count) restricted to y > 0, then the GroupNorm(1) backward with m1 = mean(g),
// m2 = mean(g * xhat) over the sample; dy is never materialised.
// Rows stream in fixed 4-row groups, two in flight, across band boundaries (the first version loaded band by band and paid one
// HBM latency per band: ~20 bands x 2 us on a p3 plane, 0.38 of the HBM peak); the pattern is recomposed at a boundary by
// wave-uniform control flow: 60 -> 50 us (stats), 84 -> 72 us (apply) HBM-cold.  Issue-bound like box_sum (20 M VALU + 13 M SALU
// per launch); splitting the big planes' rows over four waves is slower here too (66 / 92 us).
template <int VW, int MODE>
__device__ __forceinline__ void paint_rows(const BoxArgs& a, const Plane& p) {
    constexpr int G = LGD_BAND_G;
    const int lane = threadIdx.x & 63;
    const size_t base = ((size_t)p.b * a.C + p.c) * p.H * p.W;
    const float* __restrict__ src = MODE != 2 ? a.gx[p.l] + base : nullptr;
    float* __restrict__ dst = MODE != 0 ? a.out[p.l] + base : nullptr;
    const float* __restrict__ vals = a.vals + ((size_t)p.l * a.T + p.t0) * a.C + p.c;
    const int seg = p.l * a.B + p.b;
    float mu = 0.f, rs = 1.f, m1 = 0.f, m2 = 0.f;
    if (MODE != 2) { mu = a.gn_stats[2 * seg]; rs = a.gn_stats[2 * seg + 1]; }
    if (MODE == 1) { m1 = a.gn_bstats[2 * seg]; m2 = a.gn_bstats[2 * seg + 1]; }
    const bool norm = MODE != 2 || a.normalize;
    const int skip = MODE == 2 ? a.skip_last : 0;
    const int npass = (p.n + 63) >> 6;
    auto lane_val = [&](const LaneBox& bx, int pass) -> float {
        float v = 0.f;
        if (bx.x1 >= bx.x0) {  // only live boxes are read (the skipped context row may hold anything)
            v = vals[(size_t)(pass * 64 + lane) * a.C];
            if (norm) v = v / fmaxf((float)((bx.x1 - bx.x0 + 1) * (bx.y1 - bx.y0 + 1)), 1.f);
        }
        return v;
    };
    // common case (<= 64 boxes per image): rectangles and values stay in registers for the whole plane
    const LaneBox bx0 = load_lane_box(p, 0, lane, skip);
    const float val0 = lane_val(bx0, 0);
    const int bandreg = lane < p.nbp ? p.bands[lane] : p.H;
    auto band = [&](int k) {  // wave-uniform by construction
        return __builtin_amdgcn_readfirstlane(p.nbp <= 64 ? __builtin_amdgcn_readlane(bandreg, k) : (k < p.nbp ? p.bands[k] : p.H));
    };
    // MODE 0 needs only the rows some box covers (g = 0 elsewhere); the writers cover the whole plane
    int lo = 0, hi = p.H;
    if (MODE == 0) {
        int ylo = p.H, yhi = -1;
        for (int pass = 0; pass < npass; ++pass) {
            const LaneBox bx = pass ? load_lane_box(p, pass, lane, skip) : bx0;
            const bool live = bx.x1 >= bx.x0 && bx.y1 >= bx.y0;
            ylo = min(ylo, wave_min_i(live ? bx.y0 : p.H)); yhi = max(yhi, wave_max_i(live ? bx.y1 : -1));
        }
        lo = ylo; hi = max(yhi + 1, ylo);
    }
    const int ra = __builtin_amdgcn_readfirstlane(lo), rb = __builtin_amdgcn_readfirstlane(hi);
    double s1 = 0.0, s2 = 0.0;
    for (int xc = 0; rb > ra && xc < p.W; xc += 64 * VW) {
        const int xl = xc + lane * VW;
        const bool on = xl < p.W;
        float pv[VW];
        bool any = false;
        auto compose = [&](int ya) {   // the row pattern of the band that contains row ya
            #pragma unroll
            for (int j = 0; j < VW; ++j) pv[j] = 0.f;
            any = false;
            for (int pass = 0; pass < npass; ++pass) {
                LaneBox bx = bx0;
                float val = val0;
                if (pass > 0) { bx = load_lane_box(p, pass, lane, skip); val = lane_val(bx, pass); }
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
        };
        int k = 0;
        while (band(k + 1) <= ra) ++k;
        int yb = band(k + 1);
        compose(ra);
        if constexpr (MODE == 2) {
            float* col = dst + xl;
            for (int y = ra; y < rb;) {   // one row pattern per band, streamed to every row of the band
                Vec<VW> o;
                #pragma unroll
                for (int j = 0; j < VW; ++j) o.v[j] = pv[j];
                const int ye = min(yb, rb);
                if (on) for (; y < ye; ++y) vstore<VW>(col + (size_t)y * p.W, o);
                y = ye;
                if (y < rb) { ++k; yb = band(k + 1); compose(y); }
            }
        } else {
            const float* col = src + (on ? xl : 0);
            auto issue = [&](Vec<VW>* v, int y0) {
                #pragma unroll
                for (int u = 0; u < G; ++u) v[u] = vload<VW>(col + (size_t)min(y0 + u, p.H - 1) * p.W);
            };
            auto consume = [&](Vec<VW>* v, int y0) {
                #pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int y = y0 + u;
                    if (y < rb) {
                        if (y == yb) { while (band(k + 1) <= y) ++k; yb = band(k + 1); compose(y); }
                        if (on && (MODE == 1 || any)) {
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
                            else vstore<VW>(dst + xl + (size_t)y * p.W, o);
                        }
                    }
                }
            };
            Vec<VW> va[G], vb[G];
            issue(va, ra);
            for (int y0 = ra; y0 < rb; y0 += 2 * G) {
                issue(vb, y0 + G);
                consume(va, y0);
                issue(va, y0 + 2 * G);
                consume(vb, y0 + G);
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

template <int MODE>   // one wave per plane
__global__ __launch_bounds__(256) void paint_kernel(BoxArgs a) {
    const Plane p = locate(a, 4);
    if ((p.W & 3) == 0) paint_rows<4, MODE>(a, p);
    else if ((p.W & 1) == 0) paint_rows<2, MODE>(a, p);
    else paint_rows<1, MODE>(a, p);
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
    const int np = C % 8 == 0 ? 2 : 1;   // channel planes per wave
    const int nblk = lgd::fill_args(a, level_hw_host, L, B, C, T, max_n, img_off, geom, normalize, skip_last, 4 * np);
    if (nblk < 0 || !feats_host || !out) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!feats_host[l]) return LGD_EINVAL; a.in[l] = feats_host[l]; }
    a.pooled = out;
    if (T == 0) return LGD_OK;
    lgd::launch_box_sum("box_sum_kernel", a, np, nblk, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_gn_pool_fwd(const float* const* x_host, const float* gn_stats, const int32_t* level_hw_host, int L, int B, int C, int T,
                    int max_n, const int32_t* img_off, const int32_t* geom, float* out, void* stream) {
    lgd::BoxArgs a;
    const int np = C % 8 == 0 ? 2 : 1;
    const int nblk = lgd::fill_args(a, level_hw_host, L, B, C, T, max_n, img_off, geom, 1, 0, 4 * np);
    if (nblk < 0 || !x_host || !gn_stats || !out) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!x_host[l]) return LGD_EINVAL; a.in[l] = x_host[l]; }
    a.pooled = out; a.gn_stats = gn_stats;
    if (T == 0) return LGD_OK;
    lgd::launch_box_sum("gn_pool_kernel", a, np, nblk, (hipStream_t)stream);
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
    LGD_LAUNCH("gn_pool_bwd_stats_kernel", lgd::paint_kernel<0>, dim3(nblk), dim3(256), 0, s, a);
    LGD_LAUNCH("gn_pool_bwd_finalize_kernel", lgd::gn_pool_bwd_finalize_kernel, dim3(L * B), dim3(256), 0, s, a, bstats);
    LGD_LAUNCH("gn_pool_bwd_apply_kernel", lgd::paint_kernel<1>, dim3(nblk), dim3(256), 0, s, a);
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
    LGD_LAUNCH("box_paint_kernel", lgd::paint_kernel<2>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
