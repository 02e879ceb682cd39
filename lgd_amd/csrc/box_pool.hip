// K1: mask pooling (box_sum) and its fused GroupNorm(1) + ReLU form (gn_pool) as the GEMM they are in the reference.
//
// Reference arithmetic being replaced:
//   [ref: dynamic_teacher.py:95-100]   pool = mask_b (Ni,HW) @ feat_b (C,HW)^T ; / max(mask.sum(-1), 1)     (dense fp32 GEMM against a
//   materialised 0/1 mask, one call per level and image)
//
// MI355X design (HBM-bound: one pass over a pyramid, P = B*C*sum(HW)*4 bytes; measured in tools/lab/pool_mfma_lab.hip):
//   * the product runs on v_mfma_f32_16x16x4_f32 (exact fp32 fma chains): A = mask (16 boxes x 4 pixels per MFMA), B = features
//     (4 pixels x 16 channels), D = 16 boxes x 16 channels per wave.  The mask operand never exists in HBM: each workgroup generates
//     the 0/1 operand of ITS pixel chunk from the integer rectangles (box_geom.hip) into LDS once, in MFMA operand layout, and its four
//     waves (16 channels each) share it.  Before (round 2): one wave per two channel planes, lanes own columns, row bands with a
//     wave-wide prefix sum per band -- issue-bound (15-21 M VALU instructions per launch, 22-53 of 64 lanes idle per row, a p3 plane
//     a serial chain of 100 row loads): 0.55 of the HBM peak at 8 images, 0.17 at 2 images per GPU.
//   * the feature operand is streamed ONCE with fully coalesced loads -- a wave load covers 4 channel planes x 256 contiguous bytes
//     (loading in operand layout, 16 planes x 64 B per wave load, measured 3.4 TB/s against 4.6) -- and is turned into operand layout
//     through a wave-private LDS tile (row stride 68 floats: conflict-free b128 writes and reads); two 64-pixel windows per wave are
//     in flight while one is multiplied.  LDS-DMA (global_load_lds_dwordx4 with the swizzle on the source address) measured slower
//     here (44 vs 38 us): its ring costs the LDS that occupancy needs.
//   * pixels are split into chunks of CH (256 / 512) across workgroups, so that 2 images per GPU fill the chip as 8 do; a chunk's
//     16 x C partial sums go to a workspace and a second tiny launch adds them in fixed order in fp64 (bit-reproducible run to run,
//     no atomics), normalises and writes [L][T][C].
//   * gn_pool: y = relu((x - mean) * rstd) is formed in registers as rstd * max(x - mean, 0) (bit-identical: rstd > 0), pooled like
//     above, and a second product against the indicator [x > mean] (0/1 operands are exact in f16: v_mfma_f32_16x16x16_f16) counts the
//     active pixels of every (box, channel).  With R2 = sum mask * y and R1 = sum mask * [y > 0] the backward's statistics
//     mean(g), mean(g * xhat) of g = paint(dpool / count) * [y > 0] are sums over (box, channel) of dpool / count * R1 (R2): the
//     backward no longer reads x for them (round 2: a P-byte pass, 49 us at 0.47 of the HBM peak).
//   * more than 16 boxes per image: one more box tile (blockIdx -> tile) re-reads the chunk; images with fewer boxes leave at once.
#include "common.h"

namespace lgd {

typedef float pool_f4 __attribute__((ext_vector_type(4)));
typedef float pool_f4u __attribute__((ext_vector_type(4), aligned(4)));   // gfx950 runs global memory in unaligned access mode
typedef _Float16 pool_h4 __attribute__((ext_vector_type(4)));
typedef unsigned pool_u2 __attribute__((ext_vector_type(2)));

struct PoolArgs {
    const float* in[LGD_MAX_LEVELS];
    int H[LGD_MAX_LEVELS], W[LGD_MAX_LEVELS], nchunk[LGD_MAX_LEVELS];
    int blk0[LGD_MAX_LEVELS + 1];      // first block of each dispatch slot
    int lev[LGD_MAX_LEVELS];           // level handled by slot i (largest planes first)
    int partoff[LGD_MAX_LEVELS];       // first partial tile of each LEVEL
    float invW[LGD_MAX_LEVELS];
    const int32_t* img_off;
    const int32_t* geom;
    const float* gn_stats;             // [L*B][2] mean, rstd (gn_pool)
    float* part;                       // [tile][NOUT][16][C]
    float* out;                        // [L][T][C]
    float* raw;                        // gn_pool: [2][L][T][C]  R2 = sum mask * relu(xhat), R1 = sum mask * [xhat > 0]
    int L, B, C, T, max_n, ntile, normalize, skip_last;
};

typedef float pool_f2 __attribute__((ext_vector_type(2)));
// Measured and neutral (tools/pool_variants.sh, HBM-cold and inside the step: box_sum 39.3 - 40.4 us, gn_pool 41.6 - 43.9 us in every
// variant): three windows in flight per wave instead of two, the box_sum mask operand as f16 (one more resident workgroup per CU),
// 1024-pixel chunks (half the partial sums).  A pure streaming reduction of the same bytes (gn_stats) takes 37 us.
template <int GN> struct PoolA { typedef pool_f4 T; };
template <> struct PoolA<1> { typedef pool_h4 T; };   // 0/1 are exact in f16: one LDS copy serves the fp32 and the f16 product (an f32 copy
                                                      // would save 8 converts per window but costs a resident workgroup per CU: 43 -> 46 us)

// partial tile index of (level, image, box tile, chunk)
__device__ __forceinline__ int pool_tile(const PoolArgs& a, int l, int b, int tile, int chunk) {
    return a.partoff[l] + (b * a.ntile + tile) * a.nchunk[l] + chunk;
}

template <int GN, int CH>
__global__ __launch_bounds__(256) void box_pool_kernel(PoolArgs a) {
    typedef typename PoolA<GN>::T AT;
    __shared__ AT Am[CH / 16][64];          // mask operand of the chunk: step s (16 pixels), lane (box = l & 15, kg = l >> 4) -> 4 pixels
    __shared__ float Ts[4][16 * 68];        // per wave: 16 channel rows x 64 pixels (+4 pad)
    int slot = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) slot += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    const int l = a.lev[slot];
    const int idx = blockIdx.x - a.blk0[slot];
    const int ncp = (a.C + 63) >> 6;
    const int cp = idx % ncp, chunk = (idx / ncp) % a.nchunk[l], bt = idx / (ncp * a.nchunk[l]);
    const int b = bt / a.ntile, tile = bt % a.ntile;
    const int t0 = a.img_off[b], n = a.img_off[b + 1] - t0;
    if (tile * 16 >= n) return;             // this image has no boxes in this tile (its partial tiles are never read)
    const int H = a.H[l], W = a.W[l], HW = H * W;
    const int q0 = chunk * CH;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, kg = lane >> 4;
    const int npx = min(CH, HW - q0);
    const int nstep = (npx + 15) >> 4, nwf = npx >> 6;
    float mu = 0.f, rs = 1.f;
    if (GN) { mu = a.gn_stats[2 * (l * a.B + b)]; rs = a.gn_stats[2 * (l * a.B + b) + 1]; }
    const int c0 = cp * 64 + wave * 16;     // first channel of this wave
    const bool active = c0 < a.C;           // wave-uniform
    const float* img = a.in[l] + (size_t)b * a.C * HW + q0;
    float* T = Ts[wave];
    const int cs = lane >> 4, pg = lane & 15;
    // coalesced fetch: load r of a window covers channel rows 4r .. 4r+3 (lane >> 4) x 256 contiguous bytes (lane & 15)
    size_t rowoff[4];
    #pragma unroll
    for (int r = 0; r < 4; ++r) rowoff[r] = (size_t)min(c0 + 4 * r + cs, a.C - 1) * HW + 4 * pg;
    auto fetch = [&](pool_f4* v, int w) {
        #pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __builtin_nontemporal_load(reinterpret_cast<const pool_f4u*>(img + rowoff[r] + 64 * w));
    };
    pool_f4 va[4], vb[4];
    if (active && 0 < nwf) fetch(va, 0);    // in flight while the mask operand is generated
    if (active && 1 < nwf) fetch(vb, 1);
    {
        const int box = tile * 16 + m;
        int4 r = make_int4(0, -1, 0, -1);
        if (box < n && !(a.skip_last && box == n - 1)) r = reinterpret_cast<const int4*>(a.geom + geom_rects_off())[((size_t)l * a.B + b) * a.max_n + box];
        if (r.w < r.z) { r.x = 0; r.y = -1; }                      // empty in y: empty
        const float invW = a.invW[l];
        const bool aligned = ((W | q0) & 3) == 0;                  // wave-uniform: a lane's 4 pixels lie in one row
        for (int s = wave; s < nstep; s += 4) {
            AT av;
            const int q = q0 + 16 * s + 4 * kg;
            // pixel -> (row, column) by float reciprocal: exact for q < 2^23 (|error| of the quotient < 1e-4 << 0.5 / W)
            if (aligned) {
                const int y = (int)(((float)q + 0.5f) * invW), x = q - y * W;
                const bool row = y >= r.z && y <= r.w;
                #pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = (row && x + j >= r.x && x + j <= r.y) ? 1.f : 0.f;
            } else {
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = (int)(((float)(q + j) + 0.5f) * invW), x = q + j - y * W;
                    av[j] = (x >= r.x && x <= r.y && y >= r.z && y <= r.w) ? 1.f : 0.f;   // pixels past the plane: y >= H, outside every box
                }
            }
            Am[s][lane] = av;
        }
    }
    __syncthreads();
    if (!active) return;
    pool_f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, cnt = {0, 0, 0, 0};
    const pool_f2 mu2 = {mu, mu};
    auto step = [&](const AT& Am4, pool_f4 x) {
        pool_f4 A;
        #pragma unroll
        for (int j = 0; j < 4; ++j) A[j] = (float)Am4[j];
        if constexpr (GN) {
            // 79 VALU operations per 64-pixel window instead of 100 (the kernel issues next to a stream, it is not far from issue-bound):
            // packed subtract, the indicator as min(bits, 1) (x - mean >= 0 after the max: positive <=> bits > 0; exact for denormals too),
            // two indicators per register times 0x3C00 = f16 1.0
            pool_f2 lo = {x[0], x[1]}, hi = {x[2], x[3]};
            lo = lo - mu2; hi = hi - mu2;                                   // v_pk_add_f32
            x[0] = fmaxf(lo[0], 0.f); x[1] = fmaxf(lo[1], 0.f); x[2] = fmaxf(hi[0], 0.f); x[3] = fmaxf(hi[1], 0.f);
            unsigned ind[4];
            #pragma unroll
            for (int j = 0; j < 4; ++j) asm("v_min_u32 %0, 1, %1" : "=v"(ind[j]) : "v"(__float_as_uint(x[j])));
            pool_u2 pk;
            pk[0] = (unsigned)__umul24(ind[0] | (ind[1] << 16), 0x3C00u);
            pk[1] = (unsigned)__umul24(ind[2] | (ind[3] << 16), 0x3C00u);
            cnt = __builtin_amdgcn_mfma_f32_16x16x16f16(Am4, __builtin_bit_cast(pool_h4, pk), cnt, 0, 0, 0);
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0], x[0], acc0, 0, 0, 0);   // two chains: an MFMA never waits for its predecessor
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[1], x[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[2], x[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[3], x[3], acc1, 0, 0, 0);
    };
    auto work = [&](const pool_f4* v, int w) {   // a full window: branch-free
        AT A[4];
        #pragma unroll
        for (int s = 0; s < 4; ++s) A[s] = Am[4 * w + s][lane];
        #pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<pool_f4*>(&T[(4 * r + cs) * 68 + 4 * pg]) = v[r];
        pool_f4 x[4];
        #pragma unroll
        for (int s = 0; s < 4; ++s) x[s] = *reinterpret_cast<const pool_f4*>(&T[m * 68 + 16 * s + 4 * kg]);   // same wave: the LDS queue is in order
        #pragma unroll
        for (int s = 0; s < 4; ++s) step(A[s], x[s]);
    };
    for (int w = 0; w < nwf; w += 2) {
        work(va, w);
        if (w + 2 < nwf) fetch(va, w + 2);
        if (w + 1 < nwf) work(vb, w + 1);
        if (w + 3 < nwf) fetch(vb, w + 3);
    }
    if (npx & 63) {   // ragged last window of the plane: element-wise guarded loads (a value of `mean` contributes nothing)
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            pool_f4 v;
            #pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 64 * nwf + 4 * pg + j < npx ? img[rowoff[r] + 64 * nwf + j] : mu;
            *reinterpret_cast<pool_f4*>(&T[(4 * r + cs) * 68 + 4 * pg]) = v;
        }
        for (int s = 0; 4 * nwf + s < nstep; ++s) step(Am[4 * nwf + s][lane], *reinterpret_cast<const pool_f4*>(&T[m * 68 + 16 * s + 4 * kg]));
    }
    // D layout: lane holds boxes 4 kg + r (r = 0..3) of channel c0 + m
    if (c0 + m < a.C) {
        float* P = a.part + (size_t)pool_tile(a, l, b, tile, chunk) * (GN ? 2 : 1) * 16 * a.C + c0 + m;
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            P[(size_t)(4 * kg + r) * a.C] = (acc0[r] + acc1[r]) * rs;
            if (GN) P[(size_t)(16 + 4 * kg + r) * a.C] = cnt[r];
        }
    }
}

// one block per (level, box row t, 64 channels): fixed-order fp64 sum of the chunk partials, normalisation, [L][T][C].
// Wave w adds chunks w, w + 4, ... with up to 16 loads in flight per lane (a serial walk over the 33 chunks of a p3 plane was a chain
// of exposed latencies: 9 us for 6 MB), the four wave sums are added in wave order.
template <int GN>
__global__ __launch_bounds__(256) void box_pool_finalize_kernel(PoolArgs a) {
    __shared__ double red[2][4][64];
    const int l = blockIdx.x / a.T, t = blockIdx.x % a.T;
    int b = 0;
    while (b + 1 < a.B && t >= a.img_off[b + 1]) ++b;      // wave-uniform scalar walk (B is small)
    const int i = t - a.img_off[b], n = a.img_off[b + 1] - a.img_off[b];
    const int tile = i >> 4, row = i & 15;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const size_t ts = (size_t)(GN ? 2 : 1) * 16 * a.C;
    const int nc = a.nchunk[l];
    double s = 0.0, s1 = 0.0;
    if (c < a.C) {
        const float* P = a.part + (size_t)pool_tile(a, l, b, tile, 0) * ts + (size_t)row * a.C + c;
        for (int k = wave; k < nc; k += 64) {
            float v[16], v1[16];
            #pragma unroll
            for (int u = 0; u < 16; ++u) {
                v[u] = k + 4 * u < nc ? P[(k + 4 * u) * ts] : 0.f;
                v1[u] = GN && k + 4 * u < nc ? P[(k + 4 * u) * ts + 16 * (size_t)a.C] : 0.f;
            }
            #pragma unroll
            for (int u = 0; u < 16; ++u) { s += (double)v[u]; s1 += (double)v1[u]; }
        }
    }
    red[0][wave][lane] = s; red[1][wave][lane] = s1;
    __syncthreads();
    if (wave == 0 && c < a.C) {
        s = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        s1 = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        float count = 0.f;
        if (!(a.skip_last && i == n - 1)) {
            const int4 r = reinterpret_cast<const int4*>(a.geom + geom_rects_off())[((size_t)l * a.B + b) * a.max_n + i];
            if (r.y >= r.x && r.w >= r.z) count = (float)((r.y - r.x + 1) * (r.w - r.z + 1));
        }
        const float inv = (GN || a.normalize) ? fmaxf(count, 1.f) : 1.f;     // [ref: dynamic_teacher.py:97-100]
        const size_t o = ((size_t)l * a.T + t) * a.C + c;
        a.out[o] = (float)s / inv;
        if (GN) { a.raw[o] = (float)s; a.raw[(size_t)a.L * a.T * a.C + o] = (float)s1; }
    }
}

static constexpr int kPoolChunkBig = 512, kPoolChunkSmall = 256;

// chunk length: 512 pixels once that gives the chip >= ~1.2 workgroups per resident slot, else 256 (2 images per GPU)
static int pool_chunk(const int32_t* level_hw_host, int L, int B, int C, int max_n) {
    const int ntile = max_n > 16 ? (max_n + 15) / 16 : 1, ncp = (C + 63) / 64;
    auto blocks = [&](int ch) {
        long n = 0;
        for (int l = 0; l < L; ++l) n += (long)B * ntile * ncp * ((level_hw_host[2 * l] * level_hw_host[2 * l + 1] + ch - 1) / ch);
        return n;
    };
    return blocks(kPoolChunkBig) >= 1200 ? kPoolChunkBig : kPoolChunkSmall;
}

static int pool_fill(PoolArgs& a, const float* const* maps_host, const int32_t* level_hw_host, int L, int B, int C, int T, int max_n,
                     const int32_t* img_off, const int32_t* geom, int CH) {
    if (!maps_host || !level_hw_host || !img_off || !geom || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 1 || T < 0 || max_n < 0) return LGD_EINVAL;
    a.L = L; a.B = B; a.C = C; a.T = T; a.max_n = max_n; a.ntile = max_n > 16 ? (max_n + 15) / 16 : 1;
    a.img_off = img_off; a.geom = geom; a.gn_stats = nullptr; a.part = nullptr; a.out = nullptr; a.raw = nullptr;
    a.normalize = 0; a.skip_last = 0;
    const int ncp = (C + 63) / 64;
    int tiles = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.in[l] = nullptr; a.lev[l] = 0;
        a.H[l] = l < L ? level_hw_host[2 * l] : 0;
        a.W[l] = l < L ? level_hw_host[2 * l + 1] : 0;
        if (l < L && (a.H[l] < 1 || a.W[l] < 1 || (long)a.H[l] * a.W[l] >= (1L << 23) || !maps_host[l])) return LGD_EINVAL;
        a.invW[l] = l < L ? 1.0f / (float)a.W[l] : 0.f;
        a.nchunk[l] = l < L ? (a.H[l] * a.W[l] + CH - 1) / CH : 0;
        a.partoff[l] = tiles;
        tiles += B * a.ntile * a.nchunk[l];
        if (l < L) a.in[l] = maps_host[l];
    }
    // dispatch slots by DESCENDING plane size (stable): the many chunks of the big levels start first
    int order[LGD_MAX_LEVELS];
    for (int i = 0; i < L; ++i) order[i] = i;
    for (int i = 1; i < L; ++i)
        for (int j = i; j > 0 && a.H[order[j]] * a.W[order[j]] > a.H[order[j - 1]] * a.W[order[j - 1]]; --j) {
            const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
        }
    int blk = 0;
    for (int i = 0; i < LGD_MAX_LEVELS; ++i) {
        a.blk0[i] = blk;
        if (i < L) { a.lev[i] = order[i]; blk += B * a.ntile * a.nchunk[order[i]] * ncp; }
    }
    a.blk0[LGD_MAX_LEVELS] = blk;
    return blk;
}

template <int GN>
static void pool_launch(const char* name, const PoolArgs& a, int nblk, int CH, hipStream_t s) {
    if (CH == kPoolChunkBig) LGD_LAUNCH(name, (box_pool_kernel<GN, kPoolChunkBig>), dim3(nblk), dim3(256), 0, s, a);
    else LGD_LAUNCH(name, (box_pool_kernel<GN, kPoolChunkSmall>), dim3(nblk), dim3(256), 0, s, a);
    LGD_LAUNCH("box_pool_finalize_kernel", box_pool_finalize_kernel<GN>, dim3(a.L * a.T, (a.C + 63) / 64), dim3(256), 0, s, a);
}

}  // namespace lgd

extern "C" {

size_t lgd_box_pool_ws_floats(const int32_t* level_hw_host, int L, int B, int C, int max_n, int outputs) {
    if (!level_hw_host || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 1 || max_n < 0 || outputs < 1 || outputs > 2) return 0;
    const int CH = lgd::pool_chunk(level_hw_host, L, B, C, max_n);
    const int ntile = max_n > 16 ? (max_n + 15) / 16 : 1;
    size_t tiles = 0;
    for (int l = 0; l < L; ++l) tiles += (size_t)B * ntile * ((level_hw_host[2 * l] * level_hw_host[2 * l + 1] + CH - 1) / CH);
    return tiles * outputs * 16 * C;
}

int lgd_box_sum(const float* const* feats_host, const int32_t* level_hw_host, int L, int B, int C, int T, int max_n,
                const int32_t* img_off, const int32_t* geom, float* ws, float* out, int normalize, int skip_last, void* stream) {
    if (!level_hw_host || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 1 || max_n < 0) return LGD_EINVAL;
    lgd::PoolArgs a;
    const int CH = lgd::pool_chunk(level_hw_host, L, B, C, max_n);
    const int nblk = lgd::pool_fill(a, feats_host, level_hw_host, L, B, C, T, max_n, img_off, geom, CH);
    if (nblk < 0) return LGD_EINVAL;
    if (T == 0) return LGD_OK;      // no boxes at all: nothing to write (out and ws are empty)
    if (!ws || !out) return LGD_EINVAL;
    a.part = ws; a.out = out; a.normalize = normalize; a.skip_last = skip_last;
    lgd::pool_launch<0>("box_sum_kernel", a, nblk, CH, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_gn_pool_fwd(const float* const* x_host, const float* gn_stats, const int32_t* level_hw_host, int L, int B, int C, int T,
                    int max_n, const int32_t* img_off, const int32_t* geom, float* ws, float* out, float* raw, void* stream) {
    if (!level_hw_host || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 1 || max_n < 0) return LGD_EINVAL;
    lgd::PoolArgs a;
    const int CH = lgd::pool_chunk(level_hw_host, L, B, C, max_n);
    const int nblk = lgd::pool_fill(a, x_host, level_hw_host, L, B, C, T, max_n, img_off, geom, CH);
    if (nblk < 0) return LGD_EINVAL;
    if (T == 0) return LGD_OK;
    if (!gn_stats || !ws || !out || !raw) return LGD_EINVAL;
    a.part = ws; a.out = out; a.raw = raw; a.gn_stats = gn_stats; a.normalize = 1;
    lgd::pool_launch<1>("gn_pool_kernel", a, nblk, CH, (hipStream_t)stream);
    return lgd::check_launch();
}

}  // extern "C"
