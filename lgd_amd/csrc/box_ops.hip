// K3: box-paint (render fwd, mask pooling bwd) and the backward of the fused GroupNorm(1) + ReLU + mask pooling (gn_pool).
// (K1, the pooling direction -- box_sum / gn_pool forward -- lives in box_pool.hip.)
//
// Reference arithmetic being replaced (a dense fp32 GEMM against materialised 0/1 masks):
//   [ref: dynamic_teacher.py:137,173]  warp = proj^T (C,Ni) @ mask_b (Ni,HW)
//
// MI355X design (HBM-bound: one pass over a pyramid, P = B*C*sum(HW)*4 bytes):
//   * one wave64 per (level, image, channel) plane; lane l owns VW adjacent columns, so a row is
//     one coalesced 16-byte-per-lane store (VW=4 when W%4==0); the row index is wave-uniform;
//   * the masks are axis-aligned rectangles, so rows are cut into BANDS inside which the set of
//     covering boxes is constant (box_geom.hip): one row pattern is composed per band and streamed to every row of the band;
//   * lane n of the wave holds box n's rectangle / value (64 boxes per pass), band activity is a
//     single v_cmp ballot, box parameters travel by v_readlane -- no atomics, fixed order (bit-reproducible run to run).
#include "common.h"

// rows per load group of the band-streaming kernels (two groups in flight per wave and plane)
#ifndef LGD_BAND_G
#define LGD_BAND_G 4
#endif

namespace lgd {

struct BoxArgs {
    float* out[LGD_MAX_LEVELS];        // painted maps / dx
    int H[LGD_MAX_LEVELS], W[LGD_MAX_LEVELS];
    int blk0[LGD_MAX_LEVELS + 1];      // first block of each dispatch slot
    int lev[LGD_MAX_LEVELS];           // level handled by slot i (largest planes first)
    const float* vals;                 // [L][T][C]
    const int32_t* img_off;
    const int32_t* geom;
    int L, B, C, T, max_n, normalize, skip_last;
    // backward of the fused GroupNorm(1) + ReLU + pooling (gn_pool)
    const float* gn_stats;             // [L*B][2] mean, rstd
    const float* gn_bstats;            // [L*B][2] m1, m2
    const float* gx[LGD_MAX_LEVELS];   // conv output x
    const float* raw;                  // [2][L][T][C] forward sums per (box, channel): relu(xhat), [xhat > 0]
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

__device__ __forceinline__ LaneBox load_lane_box(const Plane& p, int pass, int lane, int skip_last) {
    const int n = pass * 64 + lane;
    LaneBox r{0, -1, 0, -1};
    if (n < p.n && !(skip_last && n == p.n - 1)) {
        const int4 q = reinterpret_cast<const int4*>(p.rects)[n];
        r.x0 = q.x; r.x1 = q.y; r.y0 = q.z; r.y1 = q.w;
    }
    return r;
}

// ------------------------------------------------------------------------------------------- box_paint / gn_pool backward
// One skeleton for the two kernels that apply a per-band ROW PATTERN pv[x] = sum of the values of the boxes covering (band, x):
//   MODE 2  box_paint          dst = pv                                     (render fwd, mask pooling bwd; writes only)
//   MODE 1  gn_pool bwd apply  dx = rstd * (g - m1 - xhat * m2), g = pv where x > mean        (reads x, writes dx)
// d/dx of pool(relu(GN1(x))): dy = paint(dpool / count) restricted to y > 0, then the GroupNorm(1) backward with m1 = mean(g),
// m2 = mean(g * xhat) over the sample; dy is never materialised.  m1 / m2 come from the FORWARD's per-(box, channel) sums
// (gn_pool_bwd_stats_kernel below; round 2 streamed x once more for them: MODE 0, 49 us per step).
// Rows stream in fixed 4-row groups, two in flight, across band boundaries (the first version loaded band by band and paid one
// HBM latency per band: ~20 bands x 2 us on a p3 plane, 0.38 of the HBM peak); the pattern is recomposed at a boundary by
// wave-uniform control flow: 84 -> 72 us (apply) HBM-cold.
template <int VW, int MODE>
__device__ __forceinline__ void paint_rows(const BoxArgs& a, const Plane& p, const int ra, const int rb) {
    constexpr int G = LGD_BAND_G;
    const int lane = threadIdx.x & 63;
    const size_t base = ((size_t)p.b * a.C + p.c) * p.H * p.W;
    const float* __restrict__ src = MODE != 2 ? a.gx[p.l] + base : nullptr;
    float* __restrict__ dst = a.out[p.l] + base;
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
                        if (on) {
                            Vec<VW> o;
                            #pragma unroll
                            for (int j = 0; j < VW; ++j) {
                                const float d = __fsub_rn(v[u].v[j], mu);       // the forward counted the pixels with x - mean > 0 (box_pool.hip)
                                const float g = d > 0.f ? pv[j] : 0.f;
                                o.v[j] = rs * (g - m1 - __fmul_rn(d, rs) * m2);
                            }
                            vstore<VW>(dst + xl + (size_t)y * p.W, o);
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
}

// SPLIT = 1: one wave per plane.  SPLIT = 4 (few planes: 2 images per GPU): the four waves of a workgroup take a quarter of ONE plane's
// rows each -- a wave retires a row store every ~170 ns whatever else runs, so a 100-row p3 plane is a 17 us chain that 512 planes
// cannot hide (box_paint 22 -> 12 us, gn_pool backward apply 46 -> 27 us at config 4); with 2048 planes per level the extra band
// walks cost more than the shorter chains save (66 / 92 us against 37 / 68 us at 8 images).
template <int MODE, int SPLIT>
__global__ __launch_bounds__(256) void paint_kernel(BoxArgs a) {
    const Plane p = locate(a, SPLIT == 4 ? 1 : 4);
    int ra = 0, rb = p.H;
    if (SPLIT == 4) {
        const int rows = (p.H + 3) >> 2, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        ra = min(p.H, wave * rows); rb = min(p.H, ra + rows);
    }
    if ((p.W & 3) == 0) paint_rows<4, MODE>(a, p, ra, rb);
    else if ((p.W & 1) == 0) paint_rows<2, MODE>(a, p, ra, rb);
    else paint_rows<1, MODE>(a, p, ra, rb);
}
static inline bool paint_split(int B, int C) { return (long)B * C <= 1024; }

// per (level, image): m1 = mean(g), m2 = mean(g * xhat) of g = paint(dpool / count) * [x > mean] WITHOUT a pass over x:
//   sum_px g = sum over (box, channel) of dpool / count * R1,   sum_px g * xhat = sum of dpool / count * R2,
// R1 = number of active pixels, R2 = sum of relu(xhat) of the (box, channel) from the forward (raw, box_pool.hip).
__global__ __launch_bounds__(256) void gn_pool_bwd_stats_kernel(BoxArgs a, float* bstats) {
    __shared__ double red[8];
    const int seg = blockIdx.x, l = seg / a.B, b = seg % a.B;
    const int t0 = a.img_off[b], n = a.img_off[b + 1] - t0;
    const int32_t* rects = a.geom + geom_rects_off() + ((size_t)l * a.B + b) * a.max_n * 4;
    const size_t plane = (size_t)a.L * a.T * a.C;
    double s1 = 0, s2 = 0;
    for (int i = 0; i < n; ++i) {       // wave-uniform walk over the image's boxes
        const int4 r = reinterpret_cast<const int4*>(rects)[i];
        if (r.y < r.x || r.w < r.z) continue;
        const float cnt = fmaxf((float)((r.y - r.x + 1) * (r.w - r.z + 1)), 1.f);
        const size_t row = ((size_t)l * a.T + t0 + i) * a.C;
        for (int c = threadIdx.x; c < a.C; c += 256) {
            const float v = a.vals[row + c] / cnt;      // the value the apply kernel paints
            s2 += (double)v * (double)a.raw[row + c];
            s1 += (double)v * (double)a.raw[plane + row + c];
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = s1; red[2 * (threadIdx.x >> 6) + 1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double cnt = (double)a.C * a.H[l] * a.W[l];
        bstats[2 * seg] = (float)(((red[0] + red[2]) + (red[4] + red[6])) / cnt);
        bstats[2 * seg + 1] = (float)(((red[1] + red[3]) + (red[5] + red[7])) / cnt);
    }
}

static int fill_args(BoxArgs& a, const int32_t* level_hw_host, int L, int B, int C, int T, int max_n,
                     const int32_t* img_off, const int32_t* geom, int normalize, int skip_last, int ppb) {
    if (!level_hw_host || !img_off || !geom || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 4 || (C & 3) || T < 0)
        return LGD_EINVAL;
    a.L = L; a.B = B; a.C = C; a.T = T; a.max_n = max_n; a.normalize = normalize; a.skip_last = skip_last;
    a.img_off = img_off; a.geom = geom; a.vals = nullptr;
    a.gn_stats = nullptr; a.gn_bstats = nullptr; a.raw = nullptr;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) a.gx[l] = nullptr;
    int blk = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.out[l] = nullptr;
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

int lgd_gn_pool_bwd(const float* const* x_host, const float* gn_stats, const float* dpool, const float* raw,
                    const int32_t* level_hw_host, int L, int B, int C, int T, int max_n, const int32_t* img_off, const int32_t* geom,
                    float* bstats, float* const* dx_host, void* stream) {
    lgd::BoxArgs a;
    const bool split = lgd::paint_split(B, C);
    const int nblk = lgd::fill_args(a, level_hw_host, L, B, C, T, max_n, img_off, geom, 1, 0, split ? 1 : 4);
    if (nblk < 0 || !x_host || !gn_stats || !dpool || !raw || !bstats || !dx_host) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!x_host[l] || !dx_host[l]) return LGD_EINVAL;
        a.gx[l] = x_host[l]; a.out[l] = dx_host[l];
    }
    a.vals = dpool; a.gn_stats = gn_stats; a.gn_bstats = bstats; a.raw = raw;
    hipStream_t s = (hipStream_t)stream;
    LGD_LAUNCH("gn_pool_bwd_stats_kernel", lgd::gn_pool_bwd_stats_kernel, dim3(L * B), dim3(256), 0, s, a, bstats);
    if (split) LGD_LAUNCH("gn_pool_bwd_apply_kernel", (lgd::paint_kernel<1, 4>), dim3(nblk), dim3(256), 0, s, a);
    else LGD_LAUNCH("gn_pool_bwd_apply_kernel", (lgd::paint_kernel<1, 1>), dim3(nblk), dim3(256), 0, s, a);
    return lgd::check_launch();
}

int lgd_box_paint(const float* vals, const int32_t* level_hw_host, int L, int B, int C, int T, int max_n,
                  const int32_t* img_off, const int32_t* geom, float* const* outs_host, int normalize, int skip_last,
                  void* stream) {
    lgd::BoxArgs a;
    const bool split = lgd::paint_split(B, C);
    const int nblk = lgd::fill_args(a, level_hw_host, L, B, C, T, max_n, img_off, geom, normalize, skip_last, split ? 1 : 4);
    if (nblk < 0 || !outs_host || !vals) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!outs_host[l]) return LGD_EINVAL; a.out[l] = outs_host[l]; }
    a.vals = vals;
    if (split) LGD_LAUNCH("box_paint_kernel", (lgd::paint_kernel<2, 4>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    else LGD_LAUNCH("box_paint_kernel", (lgd::paint_kernel<2, 1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
