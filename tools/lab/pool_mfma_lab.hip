// lab: mask pooling (box_sum / gn_pool; dynamic_teacher.py:81-103) as the GEMM it is in the reference -- mask (boxes x pixels) times
// feat^T (pixels x channels) on v_mfma_f32_16x16x4_f32 -- with the mask operand generated from the rectangles (never in HBM) and
// the feature operand streamed once.  Variants differ in how the B operand (16 channels x 16 pixels per MFMA group) is fetched:
//   VAR 1: lane (n = l%16 channel, kg = l/16) loads float4 of pixels 4kg..4kg+3 of plane n directly (16 planes x 64 B per wave load)
//   VAR 2: coalesced loads (4 planes x 256 B per wave load) -> wave-private LDS tile -> read back in operand layout
// hipcc --offload-arch=gfx950 -O3 -o pool_mfma_lab pool_mfma_lab.hip && ./pool_mfma_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int NL = 5;
struct Args {
    const float* in[NL];
    int H[NL], W[NL], HW[NL], nchunk[NL], blk0[NL + 1], partoff[NL];
    unsigned magic[NL];
    float invW[NL];
    const int* rects;     // [L][B][16][4] x0 x1 y0 y1 (inclusive; empty: x1 < x0)
    const float* stats;   // [L*B][2] mean, rstd
    float* part;          // [tile][NOUT][16][C]
    int L, B, C;
};

template <int GN, int VAR, int NGW, int CH, int U>
__global__ __launch_bounds__(256) void pool_kernel(Args a) {
    __shared__ h4 Am[CH / 16][64];   // 0/1 are exact in f16
    __shared__ float Ts[VAR == 2 ? 4 : 1][VAR == 2 ? 16 * 68 : 4];
    int l = 0;
    #pragma unroll
    for (int i = 1; i < NL; ++i) l += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    constexpr int NGB = 4 * NGW;
    const int idx = blockIdx.x - a.blk0[l];
    const int ncp = a.C / (16 * NGB);
    const int cp = idx % ncp, chunk = (idx / ncp) % a.nchunk[l], b = idx / (ncp * a.nchunk[l]);
    const int HW = a.HW[l], W = a.W[l];
    const unsigned magic = a.magic[l];
    const int q0 = chunk * CH;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, kg = lane >> 4;
    const int npx = min(CH, HW - q0);
    const int nstep = (npx + 15) >> 4, nfull = npx >> 4;
    {   // mask operand of this chunk: lane (m = box, kg) -> 4 pixels
        const int4 r = reinterpret_cast<const int4*>(a.rects)[(l * a.B + b) * 16 + m];
        for (int s = wave; s < nstep; s += 4) {
            h4 av;
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned q = q0 + 16 * s + 4 * kg + j;
                const int y = (int)__umulhi(q, magic), x = (int)q - y * W;
                av[j] = (x >= r.x && x <= r.y && y >= r.z && y <= r.w) ? (_Float16)1.f : (_Float16)0.f;
            }
            Am[s][lane] = av;
        }
    }
    __syncthreads();
    float mu = 0.f, rs = 1.f;
    if (GN) { mu = a.stats[2 * (l * a.B + b)]; rs = a.stats[2 * (l * a.B + b) + 1]; }
    const int c0 = cp * NGB * 16 + wave * NGW * 16;   // first channel of this wave
    const float* img = a.in[l] + ((size_t)b * a.C + c0) * HW + q0;
    f4 acc[NGW], acc1[NGW];
    #pragma unroll
    for (int g = 0; g < NGW; ++g) { acc[g] = f4{0, 0, 0, 0}; acc1[g] = f4{0, 0, 0, 0}; }
    struct AOp { h4 h; f4 f; };
    auto aop = [&](int s) { AOp o; o.h = Am[s][lane]; o.f = f4{(float)o.h[0], (float)o.h[1], (float)o.h[2], (float)o.h[3]}; return o; };
    auto mma = [&](const AOp& A, f4 x, int g) {
        if (GN) {
            h4 hb;
            #pragma unroll
            for (int j = 0; j < 4; ++j) { hb[j] = x[j] > mu ? (_Float16)1.f : (_Float16)0.f; x[j] = fmaxf((x[j] - mu) * rs, 0.f); }
            acc1[g] = __builtin_amdgcn_mfma_f32_16x16x16f16(A.h, hb, acc1[g], 0, 0, 0);
        }
        #pragma unroll
        for (int j = 0; j < 4; ++j) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A.f[j], x[j], acc[g], 0, 0, 0);
    };
    if constexpr (VAR == 1) {
        const float* p[NGW];
        #pragma unroll
        for (int g = 0; g < NGW; ++g) p[g] = img + (size_t)(g * 16 + m) * HW + 4 * kg;
        for (int s0 = 0; s0 < nfull; s0 += U) {
            f4 xv[U][NGW];
            #pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = min(s0 + u, nfull - 1);
                #pragma unroll
                for (int g = 0; g < NGW; ++g) xv[u][g] = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(p[g] + 16 * s));
            }
            #pragma unroll
            for (int u = 0; u < U; ++u) {
                if (s0 + u < nfull) {
                    const AOp A4 = aop(s0 + u);
                    #pragma unroll
                    for (int g = 0; g < NGW; ++g) mma(A4, xv[u][g], g);
                }
            }
        }
        if (nstep > nfull) {   // ragged last step of the plane: element-wise guarded
            const AOp A4 = aop(nfull);
            #pragma unroll
            for (int g = 0; g < NGW; ++g) {
                f4 x;
                #pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = (16 * nfull + 4 * kg + j < npx) ? p[g][16 * nfull + j] : mu;
                mma(A4, x, g);
            }
        }
    } else {
        // windows of 64 pixels: 4 loads per group, each 4 planes x 256 B; lane (cs = l/16, pg = l%16)
        float* T = Ts[wave];
        const int cs = lane >> 4, pg = lane & 15;
        const int nwin = (npx + 63) >> 6;
        auto fetch = [&](f4 (*v)[4], int w) {
            #pragma unroll
            for (int g = 0; g < NGW; ++g)
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int px = 64 * w + 4 * pg;
                    const float* src = img + (size_t)(g * 16 + 4 * r + cs) * HW + min(px, npx - 4);   // lab: npx >= 4; a clamped vector is masked out below
                    v[g][r] = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(src));
                    if (px + 3 >= npx) {   // ragged tail
                        #pragma unroll
                        for (int j = 0; j < 4; ++j) v[g][r][j] = px + j < npx ? src[px + j - min(px, npx - 4)] : mu;
                    }
                }
        };
        f4 va[NGW][4], vb[NGW][4];
        fetch(va, 0);
        for (int w = 0; w < nwin; w += 2) {
            if (w + 1 < nwin) fetch(vb, w + 1);
            auto work = [&](f4 (*v)[4], int ww) {
                #pragma unroll
                for (int g = 0; g < NGW; ++g) {
                    #pragma unroll
                    for (int r = 0; r < 4; ++r) *reinterpret_cast<f4*>(&T[(4 * r + cs) * 68 + 4 * pg]) = v[g][r];
                    #pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (4 * ww + s < nstep) {
                            const f4 x = *reinterpret_cast<const f4*>(&T[m * 68 + 16 * s + 4 * kg]);
                            mma(aop(4 * ww + s), x, g);
                        }
                    }
                }
            };
            work(va, w);
            if (w + 2 < nwin) fetch(va, w + 2);
            if (w + 1 < nwin) work(vb, w + 1);
        }
    }
    // partial tile of this (level, image, chunk): [NOUT][16 boxes][C]
    const int tile = a.partoff[l] + b * a.nchunk[l] + chunk;
    float* P = a.part + (size_t)tile * (GN ? 2 : 1) * 16 * a.C;
    #pragma unroll
    for (int g = 0; g < NGW; ++g)
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            P[(size_t)(4 * kg + r) * a.C + c0 + g * 16 + m] = acc[g][r];
            if (GN) P[(size_t)(16 + 4 * kg + r) * a.C + c0 + g * 16 + m] = acc1[g][r];
        }
}

// v3: coalesced loads through a wave-private LDS tile (VAR 2 of the first lab round: 4.6 TB/s vs 3.4 direct); full 64-pixel windows run
// branch-free (the ragged last window of a plane has its own path), the first windows' loads are issued BEFORE the mask operand is
// generated, mask coordinates by float reciprocal, two accumulators per group; GN: pooled = rstd * sum mask * max(x - mean, 0)
template <int GN> struct AType { typedef f4 T; };
template <> struct AType<1> { typedef h4 T; };
template <int GN, int NGW, int CH, int WPB>
__global__ __launch_bounds__(64 * WPB) void pool2_kernel(Args a) {
    typedef typename AType<GN>::T AT;
    __shared__ AT Am[CH / 16][64];
    __shared__ float Ts[WPB][16 * 68];
    int l = 0;
    #pragma unroll
    for (int i = 1; i < NL; ++i) l += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    constexpr int NGB = WPB * NGW;
    const int idx = blockIdx.x - a.blk0[l];
    const int ncp = a.C / (16 * NGB);
    const int cp = idx % ncp, chunk = (idx / ncp) % a.nchunk[l], b = idx / (ncp * a.nchunk[l]);
    const int HW = a.HW[l], W = a.W[l];
    const float invW = a.invW[l];
    const int q0 = chunk * CH;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, kg = lane >> 4;
    const int npx = min(CH, HW - q0);
    const int nstep = (npx + 15) >> 4;
    float mu = 0.f, rs = 1.f;
    if (GN) { mu = a.stats[2 * (l * a.B + b)]; rs = a.stats[2 * (l * a.B + b) + 1]; }
    const int c0 = cp * NGB * 16 + wave * NGW * 16;
    const float* img = a.in[l] + ((size_t)b * a.C + c0) * HW + q0;
    float* T = Ts[wave];
    const int cs = lane >> 4, pg = lane & 15;
    const int nwf = npx >> 6;   // full windows
    const float* lp = img + (size_t)cs * HW + 4 * pg;
    auto fetch = [&](f4 (*v)[4], int w) {
        #pragma unroll
        for (int g = 0; g < NGW; ++g)
            #pragma unroll
            for (int r = 0; r < 4; ++r) v[g][r] = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(lp + (size_t)(g * 16 + 4 * r) * HW + 64 * w));
    };
    f4 va[NGW][4], vb[NGW][4];
    if (0 < nwf) fetch(va, 0);
    if (1 < nwf) fetch(vb, 1);
    {   // mask operand: lane (m = box, kg) -> 4 pixels of step s
        const int4 r = reinterpret_cast<const int4*>(a.rects)[(l * a.B + b) * 16 + m];
        const bool aligned = ((W | q0) & 3) == 0;   // wave-uniform
        for (int s = wave; s < nstep; s += WPB) {
            AT av;
            const int q = q0 + 16 * s + 4 * kg;
            if (aligned) {
                const int y = (int)(((float)q + 0.5f) * invW), x = q - y * W;
                const bool row = y >= r.z && y <= r.w;
                #pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = (row && x + j >= r.x && x + j <= r.y) ? 1.f : 0.f;
            } else {
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = (int)(((float)(q + j) + 0.5f) * invW), x = q + j - y * W;
                    av[j] = (x >= r.x && x <= r.y && y >= r.z && y <= r.w) ? 1.f : 0.f;
                }
            }
            Am[s][lane] = av;
        }
    }
    __syncthreads();
    f4 acc[NGW][2], acc1[NGW];
    #pragma unroll
    for (int g = 0; g < NGW; ++g) { acc[g][0] = f4{0, 0, 0, 0}; acc[g][1] = f4{0, 0, 0, 0}; acc1[g] = f4{0, 0, 0, 0}; }
    auto step = [&](const AT& A, f4 x, int g) {
        f4 af;
        #pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = (float)A[j];
        if constexpr (GN) {
            unsigned ind[4];
            #pragma unroll
            for (int j = 0; j < 4; ++j) { x[j] = fmaxf(x[j] - mu, 0.f); ind[j] = min(__float_as_uint(x[j]), 1u); }
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 pk; pk[0] = (unsigned)__umul24(ind[0] | (ind[1] << 16), 0x3C00u); pk[1] = (unsigned)__umul24(ind[2] | (ind[3] << 16), 0x3C00u);   // f16 1.0 where x > mean
            acc1[g] = __builtin_amdgcn_mfma_f32_16x16x16f16(A, __builtin_bit_cast(h4, pk), acc1[g], 0, 0, 0);
        }
        #pragma unroll
        for (int j = 0; j < 4; ++j) acc[g][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], x[j], acc[g][j & 1], 0, 0, 0);
    };
    auto work = [&](f4 (*v)[4], int ww) {   // a full window
        AT A[4];
        #pragma unroll
        for (int s = 0; s < 4; ++s) A[s] = Am[4 * ww + s][lane];
        #pragma unroll
        for (int g = 0; g < NGW; ++g) {
            #pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<f4*>(&T[(4 * r + cs) * 68 + 4 * pg]) = v[g][r];
            f4 x[4];
            #pragma unroll
            for (int s = 0; s < 4; ++s) x[s] = *reinterpret_cast<const f4*>(&T[m * 68 + 16 * s + 4 * kg]);
            #pragma unroll
            for (int s = 0; s < 4; ++s) step(A[s], x[s], g);
        }
    };
    for (int w = 0; w < nwf; w += 2) {
        work(va, w);
        if (w + 2 < nwf) fetch(va, w + 2);
        if (w + 1 < nwf) work(vb, w + 1);
        if (w + 3 < nwf) fetch(vb, w + 3);
    }
    if (npx & 63) {   // ragged last window of the plane: element-wise guarded loads, steps as needed
        #pragma unroll
        for (int g = 0; g < NGW; ++g) {
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                f4 v;
                #pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = 64 * nwf + 4 * pg + j < npx ? lp[(size_t)(g * 16 + 4 * r) * HW + 64 * nwf + j] : mu;
                *reinterpret_cast<f4*>(&T[(4 * r + cs) * 68 + 4 * pg]) = v;
            }
            for (int s = 0; 4 * nwf + s < nstep; ++s) step(Am[4 * nwf + s][lane], *reinterpret_cast<const f4*>(&T[m * 68 + 16 * s + 4 * kg]), g);
        }
    }
    const int tile = a.partoff[l] + b * a.nchunk[l] + chunk;
    float* P = a.part + (size_t)tile * (GN ? 2 : 1) * 16 * a.C;
    #pragma unroll
    for (int g = 0; g < NGW; ++g)
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            P[(size_t)(4 * kg + r) * a.C + c0 + g * 16 + m] = (acc[g][0][r] + acc[g][1][r]) * rs;
            if (GN) P[(size_t)(16 + 4 * kg + r) * a.C + c0 + g * 16 + m] = acc1[g][r];
        }
}

// v4: the feature operand goes HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, D windows of 4 KB in flight per
// wave).  The DMA writes lane-linear (wave-uniform base + lane * 16), so the bank-conflict-free image is made on the SOURCE side:
// row R (channel) slot p holds pixel group (p + R) & 15; the MFMA operand read of group g is slot (g - R) & 15.
#define GLDS(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x70 | 0xF00); }
template <int GN, int CH, int WPB, int D, bool dma>
__device__ __forceinline__ void pool4_body(const Args& a, typename AType<GN>::T (*Am)[64], float (*Ring)[D][16 * 64]) {
    typedef typename AType<GN>::T AT;
    int l = 0;
    #pragma unroll
    for (int i = 1; i < NL; ++i) l += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    const int idx = blockIdx.x - a.blk0[l];
    const int ncp = a.C / (16 * WPB);
    const int cp = idx % ncp, chunk = (idx / ncp) % a.nchunk[l], b = idx / (ncp * a.nchunk[l]);
    const int HW = a.HW[l], W = a.W[l];
    const float invW = a.invW[l];
    const int q0 = chunk * CH;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, kg = lane >> 4;
    const int npx = min(CH, HW - q0);
    const int nstep = (npx + 15) >> 4;
    const int c0 = cp * WPB * 16 + wave * 16;
    const float* img = a.in[l] + ((size_t)b * a.C + c0) * HW + q0;
    const int cs = lane >> 4, pg = lane & 15;
    const int nwf = npx >> 6;                  // full windows
    auto issue = [&](int w) {
        float* dst = Ring[wave][w % D];
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int R = 4 * r + cs;
            const float* src = img + (size_t)R * HW + 64 * w + 4 * ((pg + R) & 15);
            if (dma) GLDS(src, dst + r * 256);
            else *reinterpret_cast<f4*>(dst + r * 256 + lane * 4) = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(src));
        }
    };
    #pragma unroll
    for (int w = 0; w < D; ++w) if (w < nwf) issue(w);
    float mu = 0.f, rs = 1.f;
    if (GN) { mu = a.stats[2 * (l * a.B + b)]; rs = a.stats[2 * (l * a.B + b) + 1]; }
    {
        const int4 r = reinterpret_cast<const int4*>(a.rects)[(l * a.B + b) * 16 + m];
        const bool aligned = ((W | q0) & 3) == 0;
        for (int s = wave; s < nstep; s += WPB) {
            AT av;
            const int q = q0 + 16 * s + 4 * kg;
            if (aligned) {
                const int y = (int)(((float)q + 0.5f) * invW), x = q - y * W;
                const bool row = y >= r.z && y <= r.w;
                #pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = (row && x + j >= r.x && x + j <= r.y) ? 1.f : 0.f;
            } else {
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = (int)(((float)(q + j) + 0.5f) * invW), x = q + j - y * W;
                    av[j] = (x >= r.x && x <= r.y && y >= r.z && y <= r.w) ? 1.f : 0.f;
                }
            }
            Am[s][lane] = av;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only: the DMAs stay in flight across the barrier
    __builtin_amdgcn_s_barrier();
    f4 acc[2], acc1;
    acc[0] = f4{0, 0, 0, 0}; acc[1] = f4{0, 0, 0, 0}; acc1 = f4{0, 0, 0, 0};
    auto step = [&](const AT& A, f4 x) {
        f4 af;
        #pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = (float)A[j];
        if constexpr (GN) {
            unsigned ind[4];
            #pragma unroll
            for (int j = 0; j < 4; ++j) { x[j] = fmaxf(x[j] - mu, 0.f); ind[j] = min(__float_as_uint(x[j]), 1u); }
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 pk; pk[0] = (unsigned)__umul24(ind[0] | (ind[1] << 16), 0x3C00u); pk[1] = (unsigned)__umul24(ind[2] | (ind[3] << 16), 0x3C00u);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x16f16(A, __builtin_bit_cast(h4, pk), acc1, 0, 0, 0);
        }
        #pragma unroll
        for (int j = 0; j < 4; ++j) acc[j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], x[j], acc[j & 1], 0, 0, 0);
    };
    auto work = [&](int w) {
        const float* T = Ring[wave][w % D];
        AT A[4]; f4 x[4];
        #pragma unroll
        for (int s = 0; s < 4; ++s) { A[s] = Am[4 * w + s][lane]; x[s] = *reinterpret_cast<const f4*>(&T[m * 64 + 4 * ((4 * s + kg - m) & 15)]); }
        #pragma unroll
        for (int s = 0; s < 4; ++s) step(A[s], x[s]);
    };
    int w = 0;
    for (; w + D <= nwf; ++w) {     // D windows in flight: the oldest has landed when 4 (D - 1) DMAs are outstanding
        if (dma) wait_vm<4 * (D - 1)>();
        work(w);
        if (w + D < nwf) issue(w + D);
        else { ++w; break; }
    }
    if (dma) wait_vm<0>();
    for (; w < nwf; ++w) work(w);
    if (npx & 63) {   // ragged last window of the plane
        float* T = Ring[wave][0];
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int R = 4 * r + cs, G = (pg + R) & 15;
            f4 v;
            #pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 64 * nwf + 4 * G + j < npx ? img[(size_t)R * HW + 64 * nwf + 4 * G + j] : mu;
            *reinterpret_cast<f4*>(&T[r * 256 + lane * 4]) = v;
        }
        for (int s = 0; 4 * nwf + s < nstep; ++s) step(Am[4 * nwf + s][lane], *reinterpret_cast<const f4*>(&T[m * 64 + 4 * ((4 * s + kg - m) & 15)]));
    }
    const int tile = a.partoff[l] + b * a.nchunk[l] + chunk;
    float* P = a.part + (size_t)tile * (GN ? 2 : 1) * 16 * a.C;
    #pragma unroll
    for (int r = 0; r < 4; ++r) {
        P[(size_t)(4 * kg + r) * a.C + c0 + m] = (acc[0][r] + acc[1][r]) * rs;
        if (GN) P[(size_t)(16 + 4 * kg + r) * a.C + c0 + m] = acc1[r];
    }
}
template <int GN, int CH, int WPB, int D>
__global__ __launch_bounds__(64 * WPB) void pool4_kernel(Args a) {
    __shared__ typename AType<GN>::T Am[CH / 16][64];
    __shared__ float Ring[WPB][D][16 * 64];
    int l = 0;
    #pragma unroll
    for (int i = 1; i < NL; ++i) l += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    if ((a.HW[l] & 3) == 0) pool4_body<GN, CH, WPB, D, true>(a, Am, Ring);    // 16-byte aligned rows: LDS-DMA
    else pool4_body<GN, CH, WPB, D, false>(a, Am, Ring);                      // else the windows go through registers
}

// out[l][b][16][C] (and the indicator counts) = fixed-order fp64 sum of the chunk partials
template <int GN>
__global__ void finalize_kernel(Args a, float* out, float* out1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, lb = blockIdx.z, l = lb / a.B, b = lb % a.B;
    if (c >= a.C) return;
    double s = 0, s1 = 0;
    const size_t ts = (size_t)(GN ? 2 : 1) * 16 * a.C;
    const float* P = a.part + (size_t)(a.partoff[l] + b * a.nchunk[l]) * ts + (size_t)i * a.C + c;
    const int n = a.nchunk[l];
    for (int k = 0; k < n; k += 4) {
        float v[4], v1[4];
        #pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = k + u < n ? P[(k + u) * ts] : 0.f; v1[u] = GN && k + u < n ? P[(k + u) * ts + 16 * a.C] : 0.f; }
        #pragma unroll
        for (int u = 0; u < 4; ++u) { s += v[u]; s1 += v1[u]; }
    }
    out[((size_t)lb * 16 + i) * a.C + c] = (float)s;
    if (GN) out1[((size_t)lb * 16 + i) * a.C + c] = (float)s1;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, C = 256, NBOX = 11;
    const int Hs[NL] = {100, 50, 25, 13, 7}, Ws[NL] = {168, 84, 42, 21, 11};
    const int NB = B >= 8 ? 4 : 12;   // rotating input sets (> 256 MB in total: HBM-cold)
    size_t tot = 0;
    for (int l = 0; l < NL; ++l) tot += (size_t)B * C * Hs[l] * Ws[l];
    printf("B=%d  P = %.1f MB\n", B, tot * 4 / 1e6);
    std::vector<std::vector<float*>> X(NB, std::vector<float*>(NL));
    std::vector<std::vector<float>> hx(NL);
    srand(1);
    for (int l = 0; l < NL; ++l) {
        const size_t n = (size_t)B * C * Hs[l] * Ws[l];
        hx[l].resize(n);
        for (auto& v : hx[l]) v = (rand() % 2001 - 1000) * 1e-3f;
        for (int k = 0; k < NB; ++k) { CK(hipMalloc(&X[k][l], n * 4 + 64)); CK(hipMemcpy(X[k][l], hx[l].data(), n * 4, hipMemcpyHostToDevice)); }
    }
    // rectangles: 10 random boxes + the whole image, per level by integer scaling
    std::vector<int> hr((size_t)NL * B * 16 * 4);
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < 16; ++n) {
            float x0 = rand() % 1200, y0 = rand() % 700, w = 20 + rand() % 500, h = 20 + rand() % 400;
            if (n == NBOX - 1) { x0 = 0; y0 = 0; w = 1344; h = 800; }
            for (int l = 0; l < NL; ++l) {
                const float s = 8 << l;
                int* r = &hr[(((size_t)l * B + b) * 16 + n) * 4];
                r[0] = (int)(x0 / s); r[1] = std::min(Ws[l] - 1, (int)((x0 + w) / s)); r[2] = (int)(y0 / s); r[3] = std::min(Hs[l] - 1, (int)((y0 + h) / s));
                if (n >= NBOX) { r[0] = 0; r[1] = -1; r[2] = 0; r[3] = -1; }
            }
        }
    int* drects; CK(hipMalloc(&drects, hr.size() * 4)); CK(hipMemcpy(drects, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hs(NL * B * 2);
    for (int i = 0; i < NL * B; ++i) { hs[2 * i] = 0.05f * (i % 5 - 2); hs[2 * i + 1] = 1.7f + 0.01f * i; }
    float* dstats; CK(hipMalloc(&dstats, hs.size() * 4)); CK(hipMemcpy(dstats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    float *dout, *dout1, *dpart;
    CK(hipMalloc(&dout, (size_t)NL * B * 16 * C * 4)); CK(hipMalloc(&dout1, (size_t)NL * B * 16 * C * 4));
    CK(hipMalloc(&dpart, (size_t)64 << 20));
    // CPU reference (fp64)
    std::vector<double> ref[2], ref1;
    for (int gn = 0; gn < 2; ++gn) ref[gn].assign((size_t)NL * B * 16 * C, 0.0);
    ref1.assign((size_t)NL * B * 16 * C, 0.0);
    for (int l = 0; l < NL; ++l)
        for (int b = 0; b < B; ++b)
            for (int n = 0; n < NBOX; ++n) {
                const int* r = &hr[(((size_t)l * B + b) * 16 + n) * 4];
                const float mu = hs[2 * (l * B + b)], rs = hs[2 * (l * B + b) + 1];
                for (int c = 0; c < C; ++c) {
                    const float* p = &hx[l][((size_t)b * C + c) * Hs[l] * Ws[l]];
                    double s = 0, sg = 0, s1 = 0;
                    for (int y = r[2]; y <= r[3]; ++y)
                        for (int x = r[0]; x <= r[1]; ++x) {
                            const float v = p[y * Ws[l] + x];
                            s += v; sg += fmaxf((v - mu) * rs, 0.f); s1 += v > mu;
                        }
                    const size_t o = (((size_t)l * B + b) * 16 + n) * C + c;
                    ref[0][o] = s; ref[1][o] = sg; ref1[o] = s1;
                }
            }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int gn, int CH, int NGW, auto launch, int WPB = 4) {
        Args a{};
        a.L = NL; a.B = B; a.C = C; a.rects = drects; a.stats = dstats; a.part = dpart;
        int blk = 0, tiles = 0;
        for (int l = 0; l < NL; ++l) {
            a.H[l] = Hs[l]; a.W[l] = Ws[l]; a.HW[l] = Hs[l] * Ws[l];
            a.magic[l] = (unsigned)((0x100000000ull + Ws[l] - 1) / Ws[l]); a.invW[l] = 1.0f / Ws[l];
            a.nchunk[l] = (a.HW[l] + CH - 1) / CH;
            a.blk0[l] = blk; blk += B * a.nchunk[l] * (C / (16 * WPB * NGW));
            a.partoff[l] = tiles; tiles += B * a.nchunk[l];
        }
        a.blk0[NL] = blk;
        auto go = [&](int k) {
            for (int l = 0; l < NL; ++l) a.in[l] = X[k % NB][l];
            launch(a, blk);
        };
        for (int i = 0; i < 3; ++i) go(i);
        CK(hipDeviceSynchronize());
        float tot_ms = 0, best = 1e9, fin = 0;
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e0)); go(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot_ms += ms; best = fminf(best, ms);
            CK(hipEventRecord(e0));
            if (gn) finalize_kernel<1><<<dim3((C + 255) / 256, 16, NL * B), 256>>>(a, dout, dout1);
            else finalize_kernel<0><<<dim3((C + 255) / 256, 16, NL * B), 256>>>(a, dout, dout1);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); fin += ms;
        }
        CK(hipGetLastError());
        std::vector<float> ho((size_t)NL * B * 16 * C), ho1(ho.size());
        CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ho1.data(), dout1, ho.size() * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0, e1max = 0;
        for (size_t i = 0; i < ho.size(); ++i) {
            const double d = ho[i] - ref[gn][i]; num += d * d; den += ref[gn][i] * ref[gn][i];
            if (gn) e1max = fmax(e1max, fabs(ho1[i] - ref1[i]));
        }
        printf("%-44s blocks %5d  avg %6.1f us %5.0f GB/s  best %6.1f us  finalize %5.1f us  partials %5.2f MB  rel err %.1e  count err %.0f\n", name, blk,
               tot_ms / 12 * 1e3, tot * 4 / (tot_ms / 12 * 1e-3) / 1e9, best * 1e3, fin / 12 * 1e3, tiles * (gn ? 2 : 1) * 16.0 * C * 4 / 1e6, sqrt(num / den), e1max);
    };
#define R(GN, VAR, NGW, CH, U) run("v1 GN=" #GN " VAR=" #VAR " NGW=" #NGW " CH=" #CH " U=" #U, GN, CH, NGW, [&](const Args& a, int blk) { pool_kernel<GN, VAR, NGW, CH, U><<<blk, 256>>>(a); });
#define R2(GN, NGW, CH, WPB) run("v2 GN=" #GN " NGW=" #NGW " CH=" #CH " WPB=" #WPB, GN, CH, NGW, [&](const Args& a, int blk) { pool2_kernel<GN, NGW, CH, WPB><<<blk, 64 * WPB>>>(a); }, WPB);
#define R4(GN, CH, WPB, D) run("v4 GN=" #GN " CH=" #CH " WPB=" #WPB " D=" #D, GN, CH, 1, [&](const Args& a, int blk) { pool4_kernel<GN, CH, WPB, D><<<blk, 64 * WPB>>>(a); }, WPB);
    R2(0, 1, 512, 4) R2(1, 1, 512, 8)
    R4(0, 512, 4, 2) R4(0, 512, 4, 3) R4(0, 512, 4, 4) R4(0, 512, 8, 3) R4(0, 1024, 4, 3) R4(0, 1024, 8, 3) R4(0, 256, 4, 3) R4(0, 512, 2, 4)
    R4(1, 512, 4, 2) R4(1, 512, 4, 3) R4(1, 512, 4, 4) R4(1, 512, 8, 3) R4(1, 1024, 4, 3) R4(1, 1024, 8, 3) R4(1, 256, 4, 3) R4(1, 512, 2, 4)
    return 0;
}
