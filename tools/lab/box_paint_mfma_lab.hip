// LAB (not built into the library): the painting direction of K1/K3 as the transposed MFMA product of csrc/box_pool.hip -- second version
// (a wave owns every fourth 64-pixel window and walks four 16-channel groups with the mask operand in registers; the first version
// shared the chunk's operand through LDS like box_pool.hip).  Both pass tests/test_kernels_gpu.py (mask_pool / render_paint / gn_relu) when
// dropped into csrc/ in place of box_ops.hip's entry points.  Measured HBM-cold (tools/kbench.py), box_paint / gn_pool backward apply:
//   8 images: 44.4 / 75.3 us (LDS operand), 47.5 / 92.4 us (this file)  vs  36.3 / 68.8 us for the band kernels of csrc/box_ops.hip
//   2 images: 17.8 / 24.7 us (LDS operand), 18.3 / 32.4 us (this file)  vs  19.8 / 28.8 us for the band kernels with rows split over 4 waves
// A painted row band is ONE pattern streamed to every row of the band: a store per 16 bytes and nothing else; the product spends four
// MFMAs and an LDS round trip on the same 16 bytes.  Rejected for the product.
// K3: box-paint (render fwd, mask pooling bwd) and the backward of the fused GroupNorm(1) + ReLU + mask pooling (gn_pool),
// as the transposed form of the MFMA product of box_pool.hip.
//
// Reference arithmetic being replaced (a dense fp32 GEMM against materialised 0/1 masks):
//   [ref: dynamic_teacher.py:137,173]  warp = proj^T (C,Ni) @ mask_b (Ni,HW)        (and autograd's transpose of 95-100)
//
// MI355X design (HBM-bound: P bytes written, for the gn_pool backward P read + P written):
//   * painted[c][px] = sum_boxes val[box][c] * mask[box][px] on v_mfma_f32_16x16x4_f32 with K = boxes: A = mask^T (16 pixels x 4 boxes
//     per MFMA, generated from the integer rectangles in registers once per 64-pixel window and used for four 16-channel groups),
//     B = the boxes' values (4 boxes x 16 channels, held in registers), D = 16 pixels x 16 channels: a lane ends up
//     with 4 consecutive pixels of one channel plane -- the operand layout of box_pool.hip -- and the same wave-private padded LDS tile
//     turns a 64-pixel window around so that every global store covers 4 planes x 256 contiguous bytes.
//   * pixels are split into chunks across workgroups like the pooling direction (round 2: one wave per channel plane, lanes own columns,
//     one row pattern per row band composed by a loop over the active boxes: a p3 plane was a serial chain of 100 row stores, which at 2
//     images per GPU left 512 long waves for 256 CUs -- 2.0 TB/s).
//   * gn_pool backward apply: x streams in through the same tile, dx = rstd * (g - m1 - xhat * m2) with g = painted where x > mean is
//     formed in operand layout and leaves through the tile again; m1 / m2 come from the forward's per-(box, channel) sums
//     (gn_pool_bwd_stats_kernel: no pass over x).
//   * more than 16 boxes per image: further 16-box tiles accumulate into the same registers (their rectangles / values are re-loaded
//     per window: L2 hits; the usual <= 16 boxes keep them in registers for the whole chunk).
#include "common.h"

namespace lgd {

typedef float paint_f4 __attribute__((ext_vector_type(4)));
typedef float paint_f4u __attribute__((ext_vector_type(4), aligned(4)));

struct PaintArgs {
    float* out[LGD_MAX_LEVELS];        // painted maps / dx
    const float* gx[LGD_MAX_LEVELS];   // gn_pool backward: conv output x
    int H[LGD_MAX_LEVELS], W[LGD_MAX_LEVELS], nchunk[LGD_MAX_LEVELS];
    int blk0[LGD_MAX_LEVELS + 1];      // first block of each dispatch slot
    int lev[LGD_MAX_LEVELS];           // level handled by slot i (largest planes first)
    float invW[LGD_MAX_LEVELS];
    const float* vals;                 // [L][T][C]
    const float* gn_stats;             // [L*B][2] mean, rstd
    const float* gn_bstats;            // [L*B][2] m1, m2
    const float* raw;                  // [2][L][T][C] forward sums per (box, channel): relu(xhat), [xhat > 0]
    const int32_t* img_off;
    const int32_t* geom;
    int L, B, C, T, max_n, normalize, skip_last;
};

// MODE 2  box_paint          dst = painted                                  (render fwd, mask pooling bwd; writes only)
// MODE 1  gn_pool bwd apply  dx = rstd * (g - m1 - xhat * m2), g = painted where x > mean        (reads x, writes dx)
// A workgroup owns (level, image, pixel chunk, 64 channels); each of its four waves owns every fourth 64-pixel window of the chunk and
// walks the four 16-channel groups with it: the window's mask operand is generated once in registers and serves 4 x 16 MFMAs -- no
// shared operand, no barrier, nothing between a wave's start and its first store but one round of loads (rectangles + values).  (First
// version: the chunk's operand in LDS shared by four waves that each own 16 channels, as in box_pool.hip -- correct, and 44 us where
// the round-2 band kernel took 36: half of a workgroup's life was the prologue in front of the barrier, with nothing to overlap it.)
template <int MODE, int CH>
__global__ __launch_bounds__(256) void box_paint_kernel(PaintArgs a) {
    __shared__ float Ts[4][16 * 68];           // per wave: 16 channel rows x 64 pixels (+4 pad)
    int slot = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) slot += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    const int l = a.lev[slot];
    const int idx = blockIdx.x - a.blk0[slot];
    const int ncp = (a.C + 63) >> 6;
    const int cp = idx % ncp, chunk = (idx / ncp) % a.nchunk[l], b = idx / (ncp * a.nchunk[l]);
    const int t0 = a.img_off[b], n = a.img_off[b + 1] - t0;
    const int H = a.H[l], W = a.W[l], HW = H * W;
    const int q0 = chunk * CH;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, kg = lane >> 4;      // operand / D layout: channel m, pixels 4 kg .. 4 kg + 3 of a 16-pixel step
    const int cs = lane >> 4, pg = lane & 15;     // coalesced layout: channel row 4 r + cs, pixels 4 pg .. 4 pg + 3 of a 64-pixel window
    const int npx = min(CH, HW - q0);
    const int nwin = (npx + 63) >> 6;
    const int c0 = cp * 64;
    const int seg = l * a.B + b;
    float mu = 0.f, rs = 1.f, m1 = 0.f, m2 = 0.f;
    if (MODE == 1) { mu = a.gn_stats[2 * seg]; rs = a.gn_stats[2 * seg + 1]; m1 = a.gn_bstats[2 * seg]; m2 = a.gn_bstats[2 * seg + 1]; }
    const bool norm = MODE == 1 || a.normalize;
    const int skip = MODE == 2 ? a.skip_last : 0;
    const int4* rects = reinterpret_cast<const int4*>(a.geom + geom_rects_off()) + ((size_t)l * a.B + b) * a.max_n;
    const float* vals = a.vals + ((size_t)l * a.T + t0) * a.C;
    float* T = Ts[wave];
    const float invW = a.invW[l];
    const int ntiles = max(1, (n + 15) >> 4);
    // lane (pixel = l & 15, kq = l >> 4) <-> boxes 4 ks + kq of a 16-box tile (A operand); lane (kq, channel = l & 15) <-> their values (B)
    int4 rc[4];
    float bv[4][4];
    auto load_tile = [&](int t) {
        float raw[4][4];
        #pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int box = t * 16 + 4 * ks + kg;
            const bool in = box < n && !(skip && box == n - 1);
            rc[ks] = in ? rects[box] : make_int4(0, -1, 0, -1);
            #pragma unroll
            for (int g = 0; g < 4; ++g) raw[g][ks] = (in && c0 + 16 * g + m < a.C) ? vals[(size_t)box * a.C + c0 + 16 * g + m] : 0.f;   // issued with the rectangles
        }
        #pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bool live = rc[ks].y >= rc[ks].x && rc[ks].w >= rc[ks].z;
            if (!live) { rc[ks].x = 0; rc[ks].y = -1; }
            const float inv = norm ? fmaxf((float)((rc[ks].y - rc[ks].x + 1) * (rc[ks].w - rc[ks].z + 1)), 1.f) : 1.f;
            #pragma unroll
            for (int g = 0; g < 4; ++g) bv[g][ks] = live ? (norm ? raw[g][ks] / inv : raw[g][ks]) : 0.f;   // a dead box contributes nothing, whatever its row holds
        }
    };
    if (ntiles == 1) load_tile(0);
    for (int w = wave; w < nwin; w += 4) {
        const bool full = 64 * w + 64 <= npx;
        paint_f4 acc[4][4];
        #pragma unroll
        for (int g = 0; g < 4; ++g)
            #pragma unroll
            for (int s = 0; s < 4; ++s) acc[g][s] = paint_f4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < ntiles; ++t) {
            if (ntiles > 1) load_tile(t);
            #pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int q = q0 + 64 * w + 16 * s + m;
                const int y = (int)(((float)q + 0.5f) * invW), x = q - y * W;   // exact below 2^23 pixels (box_pool.hip); pixels past the plane: y >= H
                float A[4];
                #pragma unroll
                for (int ks = 0; ks < 4; ++ks) A[ks] = (x >= rc[ks].x && x <= rc[ks].y && y >= rc[ks].z && y <= rc[ks].w) ? 1.f : 0.f;
                #pragma unroll
                for (int g = 0; g < 4; ++g)
                    #pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc[g][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ks], bv[g][ks], acc[g][s], 0, 0, 0);
            }
        }
        // acc[g][s][r] = painted value at pixel 16 s + 4 kg + r of channel c0 + 16 g + m
        #pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (c0 + 16 * g >= a.C) break;   // wave-uniform
            size_t rowoff[4];
            bool rowok[4];
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + 16 * g + 4 * r + cs;
                rowok[r] = c < a.C;
                rowoff[r] = ((size_t)b * a.C + min(c, a.C - 1)) * HW + q0 + 64 * w + 4 * pg;
            }
            if (MODE == 1) {
                paint_f4 xv[4];
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (full) xv[r] = __builtin_nontemporal_load(reinterpret_cast<const paint_f4u*>(a.gx[l] + rowoff[r]));
                    else {
                        #pragma unroll
                        for (int j = 0; j < 4; ++j) xv[r][j] = 64 * w + 4 * pg + j < npx ? a.gx[l][rowoff[r] + j] : mu;
                    }
                }
                #pragma unroll
                for (int r = 0; r < 4; ++r) *reinterpret_cast<paint_f4*>(&T[(4 * r + cs) * 68 + 4 * pg]) = xv[r];
                #pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const paint_f4 xo = *reinterpret_cast<const paint_f4*>(&T[m * 68 + 16 * s + 4 * kg]);   // same wave: the LDS queue is in order
                    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = __fsub_rn(xo[j], mu);       // the forward counted the pixels with x - mean > 0 (box_pool.hip)
                        const float gg = d > 0.f ? acc[g][s][j] : 0.f;
                        acc[g][s][j] = rs * (gg - m1 - __fmul_rn(d, rs) * m2);
                    }
                }
            }
            #pragma unroll
            for (int s = 0; s < 4; ++s) *reinterpret_cast<paint_f4*>(&T[m * 68 + 16 * s + 4 * kg]) = acc[g][s];
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const paint_f4 o = *reinterpret_cast<const paint_f4*>(&T[(4 * r + cs) * 68 + 4 * pg]);
                if (!rowok[r]) continue;
                float* dst = a.out[l] + rowoff[r];
                if (full) *reinterpret_cast<paint_f4u*>(dst) = o;
                else {
                    #pragma unroll
                    for (int j = 0; j < 4; ++j) if (64 * w + 4 * pg + j < npx) dst[j] = o[j];
                }
            }
        }
    }
}

// per (level, image): m1 = mean(g), m2 = mean(g * xhat) of g = paint(dpool / count) * [x > mean] WITHOUT a pass over x:
//   sum_px g = sum over (box, channel) of dpool / count * R1,   sum_px g * xhat = sum of dpool / count * R2,
// R1 = number of active pixels, R2 = sum of relu(xhat) of the (box, channel) from the forward (raw, box_pool.hip).
__global__ __launch_bounds__(256) void gn_pool_bwd_stats_kernel(PaintArgs a, float* bstats) {
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

#ifndef LGD_PAINT_CHUNK_BIG
#define LGD_PAINT_CHUNK_BIG 512
#endif
static constexpr int kPaintChunkBig = LGD_PAINT_CHUNK_BIG, kPaintChunkSmall = 256;

// chunk length: the big one once that gives the chip >= ~1.2 workgroups per resident slot, else 256 (2 images per GPU)
static int paint_chunk(const int32_t* level_hw_host, int L, int B, int C) {
    long blocks = 0;
    for (int l = 0; l < L; ++l) blocks += (long)B * ((C + 63) / 64) * ((level_hw_host[2 * l] * level_hw_host[2 * l + 1] + kPaintChunkBig - 1) / kPaintChunkBig);
    return blocks >= 1200 ? kPaintChunkBig : kPaintChunkSmall;
}

static int paint_fill(PaintArgs& a, const int32_t* level_hw_host, int L, int B, int C, int T, int max_n, const int32_t* img_off,
                      const int32_t* geom, int normalize, int skip_last, int CH) {
    if (!level_hw_host || !img_off || !geom || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 1 || T < 0 || max_n < 0) return LGD_EINVAL;
    a.L = L; a.B = B; a.C = C; a.T = T; a.max_n = max_n; a.normalize = normalize; a.skip_last = skip_last;
    a.img_off = img_off; a.geom = geom; a.vals = nullptr; a.gn_stats = nullptr; a.gn_bstats = nullptr; a.raw = nullptr;
    const int ncp = (C + 63) / 64;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.out[l] = nullptr; a.gx[l] = nullptr; a.lev[l] = 0;
        a.H[l] = l < L ? level_hw_host[2 * l] : 0;
        a.W[l] = l < L ? level_hw_host[2 * l + 1] : 0;
        if (l < L && (a.H[l] < 1 || a.W[l] < 1 || (long)a.H[l] * a.W[l] >= (1L << 23))) return LGD_EINVAL;
        a.invW[l] = l < L ? 1.0f / (float)a.W[l] : 0.f;
        a.nchunk[l] = l < L ? (a.H[l] * a.W[l] + CH - 1) / CH : 0;
    }
    // dispatch slots by DESCENDING plane size (stable)
    int order[LGD_MAX_LEVELS];
    for (int i = 0; i < L; ++i) order[i] = i;
    for (int i = 1; i < L; ++i)
        for (int j = i; j > 0 && a.H[order[j]] * a.W[order[j]] > a.H[order[j - 1]] * a.W[order[j - 1]]; --j) {
            const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
        }
    int blk = 0;
    for (int i = 0; i < LGD_MAX_LEVELS; ++i) {
        a.blk0[i] = blk;
        if (i < L) { a.lev[i] = order[i]; blk += B * a.nchunk[order[i]] * ncp; }
    }
    a.blk0[LGD_MAX_LEVELS] = blk;
    return blk;
}

template <int MODE>
static void paint_launch(const char* name, const PaintArgs& a, int nblk, int CH, hipStream_t s) {
    if (CH == kPaintChunkBig) LGD_LAUNCH(name, (box_paint_kernel<MODE, kPaintChunkBig>), dim3(nblk), dim3(256), 0, s, a);
    else LGD_LAUNCH(name, (box_paint_kernel<MODE, kPaintChunkSmall>), dim3(nblk), dim3(256), 0, s, a);
}

}  // namespace lgd

extern "C" {

int lgd_box_paint(const float* vals, const int32_t* level_hw_host, int L, int B, int C, int T, int max_n,
                  const int32_t* img_off, const int32_t* geom, float* const* outs_host, int normalize, int skip_last,
                  void* stream) {
    if (!level_hw_host || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 1) return LGD_EINVAL;
    lgd::PaintArgs a;
    const int CH = lgd::paint_chunk(level_hw_host, L, B, C);
    const int nblk = lgd::paint_fill(a, level_hw_host, L, B, C, T, max_n, img_off, geom, normalize, skip_last, CH);
    if (nblk < 0 || !outs_host || !vals) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) { if (!outs_host[l]) return LGD_EINVAL; a.out[l] = outs_host[l]; }
    a.vals = vals;
    lgd::paint_launch<2>("box_paint_kernel", a, nblk, CH, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_gn_pool_bwd(const float* const* x_host, const float* gn_stats, const float* dpool, const float* raw,
                    const int32_t* level_hw_host, int L, int B, int C, int T, int max_n, const int32_t* img_off, const int32_t* geom,
                    float* bstats, float* const* dx_host, void* stream) {
    if (!level_hw_host || L < 1 || L > LGD_MAX_LEVELS || B < 1 || C < 1) return LGD_EINVAL;
    lgd::PaintArgs a;
    const int CH = lgd::paint_chunk(level_hw_host, L, B, C);
    const int nblk = lgd::paint_fill(a, level_hw_host, L, B, C, T, max_n, img_off, geom, 1, 0, CH);
    if (nblk < 0 || !x_host || !gn_stats || !dpool || !raw || !bstats || !dx_host) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!x_host[l] || !dx_host[l]) return LGD_EINVAL;
        a.gx[l] = x_host[l]; a.out[l] = dx_host[l];
    }
    a.vals = dpool; a.gn_stats = gn_stats; a.gn_bstats = bstats; a.raw = raw;
    hipStream_t s = (hipStream_t)stream;
    LGD_LAUNCH("gn_pool_bwd_stats_kernel", lgd::gn_pool_bwd_stats_kernel, dim3(L * B), dim3(256), 0, s, a, bstats);
    lgd::paint_launch<1>("gn_pool_bwd_apply_kernel", a, nblk, CH, s);
    return lgd::check_launch();
}

}  // extern "C"
