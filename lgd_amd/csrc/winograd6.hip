// Winograd F(6x6, 3x3) data transforms: 8x8 input windows at stride 6, 64 frequencies, 6x6 outputs per tile
// (interpolation points 0, +-1, +-2, +-1/2, inf).  Same pipeline, layouts and fusions as the F(4x4,3x3) kernels of winograd.hip
//   [ref: every nn.Conv2d(., ., 3, padding=1) of the path -- dynamic_teacher.py:57,61,67-73, sequential_convs.py:10-12, the head
//    towers run on student AND teacher features distillator.py:107-109 -> retinanet.py:36-43 / thirdparty_heads/fcos.py:455-470]
// with 64 / 36 = 1.78 multiplies per output pixel instead of 36 / 16 = 2.25: the per-frequency channel GEMMs (70 % of the step) and
// the frequency buffers ([C][64][T], 1.78x the activations instead of 2.25x) both shrink to 0.79x.
// fp32 conditioning (tools/lab/wino_f6_numerics.py, every stage rounded to fp32, 256 channels, against the fp64 direct convolution):
// max error 1.8e-5 of the output scale, rms 3.8e-6 -- 1.8x F(4x4,3x3) (1.05e-5 / 2.1e-6, which is what the F(4x4) kernels measure on the
// GPU), far inside the path's 1e-4 bar (F(4x4): teacher features 1-3e-6 from the reference).
//   B^T = [1 0 -21/4 0 21/4 0 -1 0; 0 1 1 -17/4 -17/4 1 1 0; 0 -1 1 17/4 -17/4 -1 1 0; 0 1/2 1/4 -5/2 -5/4 2 1 0;
//          0 -1/2 1/4 5/2 -5/4 -2 1 0; 0 2 4 -5/2 -5 1/2 1 0; 0 -2 4 5/2 -5 -1/2 1 0; 0 -1 0 21/4 0 -21/4 0 1]
//   A^T = [1 1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 -1/2 0; 0 1 1 4 4 1/4 1/4 0; 0 1 -1 8 -8 1/8 -1/8 0; 0 1 1 16 16 1/16 1/16 0;
//          0 1 -1 32 -32 1/32 -1/32 1]
//   G   = [1 0 0; -2/9 -2/9 -2/9; -2/9 2/9 -2/9; 1/90 1/45 2/45; 1/90 -1/45 2/45; 32/45 16/45 8/45; 32/45 -16/45 8/45; 0 0 1]
// One thread per tile.  A tile's own 6 columns start at the even column 6 tx: with W % 4 == 0 they are one aligned float4 and one
// aligned float2 per row (float4 first for even tx, float2 first for odd tx -- a per-lane select), the two halo columns are the
// neighbour lanes' edge values (one DPP move each); the last tile of a row may overhang the map (W % 6 != 0): a vector lies either
// fully inside or fully outside because W % 4 == 0.  The 64 values of the workgroup's 256 tiles go through LDS two frequency rows
// (16 planes, 16 KB) at a time so that every plane is written / read as ONE 1 KB run.
// ReLU / activation masks: 36 bits per tile in a uint64 table [C][T], bit 6 i + j = pixel (i, j) of the tile's own block.
#include "winograd.h"

// minimum waves per SIMD the register allocator is held to, per kernel (lab knobs: tools/wino6_variants.sh measures them)
#ifndef LGD_W6_IN_WAVES
#define LGD_W6_IN_WAVES 3
#endif
#ifndef LGD_W6_OUT_WAVES
#define LGD_W6_OUT_WAVES 3
#endif
#ifndef LGD_W6_OUTT_WAVES
#define LGD_W6_OUTT_WAVES 5
#endif
#ifndef LGD_W6_INT_WAVES
#define LGD_W6_INT_WAVES 3
#endif

namespace lgd {

// B^T d: 8 -> 8
__device__ __forceinline__ void bt8(const float* d, float* t) {
    const float a = d[2] + d[6] - 4.25f * d[4], b = d[1] + d[5] - 4.25f * d[3];
    const float c = d[6] + 0.25f * d[2] - 1.25f * d[4], e = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
    const float f = d[6] + 4.f * d[2] - 5.f * d[4], g = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
    t[0] = (d[0] - d[6]) + 5.25f * (d[4] - d[2]);
    t[1] = a + b; t[2] = a - b; t[3] = c + e; t[4] = c - e; t[5] = f + g; t[6] = f - g;
    t[7] = (d[7] - d[1]) + 5.25f * (d[3] - d[5]);
}
// A^T m: 8 -> 6
__device__ __forceinline__ void at8(const float* m, float* y) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
    y[0] = m[0] + s12 + s34 + s56;
    y[1] = d12 + 2.f * d34 + 0.5f * d56;
    y[2] = s12 + 4.f * s34 + 0.25f * s56;
    y[3] = d12 + 8.f * d34 + 0.125f * d56;
    y[4] = s12 + 16.f * s34 + 0.0625f * s56;
    y[5] = d12 + 32.f * d34 + 0.03125f * d56 + m[7];
}
// A g: 6 -> 8 (adjoint of at8)
__device__ __forceinline__ void a8(const float* g, float* r) {
    const float e1 = g[0] + g[2] + g[4], o1 = g[1] + g[3] + g[5];
    const float e2 = g[0] + 4.f * g[2] + 16.f * g[4], o2 = 2.f * g[1] + 8.f * g[3] + 32.f * g[5];
    const float e3 = g[0] + 0.25f * g[2] + 0.0625f * g[4], o3 = 0.5f * g[1] + 0.125f * g[3] + 0.03125f * g[5];
    r[0] = g[0]; r[1] = e1 + o1; r[2] = e1 - o1; r[3] = e2 + o2; r[4] = e2 - o2; r[5] = e3 + o3; r[6] = e3 - o3; r[7] = g[5];
}
// z = B g (adjoint of bt8), entries 1..6 (the rows / columns inside the tile's own block); z0 = g0 and z7 = g7 belong to the neighbours
__device__ __forceinline__ void b8mid(const float* g, float* z) {
    const float d12 = g[1] - g[2], s12 = g[1] + g[2], d34 = g[3] - g[4], s34 = g[3] + g[4], d56 = g[5] - g[6], s56 = g[5] + g[6];
    z[0] = d12 + 0.5f * d34 + 2.f * d56 - g[7];
    z[1] = -5.25f * g[0] + s12 + 0.25f * s34 + 4.f * s56;
    z[2] = -4.25f * d12 - 2.5f * (d34 + d56) + 5.25f * g[7];
    z[3] = 5.25f * g[0] - 4.25f * s12 - 1.25f * s34 - 5.f * s56;
    z[4] = d12 + 2.f * d34 + 0.5f * d56 - 5.25f * g[7];
    z[5] = s12 + s34 + s56 - g[0];
}
// rows (2 PH, 2 PH + 1) of g added into z[0..5] = (B g)[1..6]: the frequency rows arrive two at a time (one LDS phase), so the
// column pass accumulates instead of holding all 64 values
template <int PH>
__device__ __forceinline__ void b8acc(float ga, float gb, float* z) {
    if constexpr (PH == 0) { z[0] = gb; z[1] = gb - 5.25f * ga; z[2] = -4.25f * gb; z[3] = 5.25f * ga - 4.25f * gb; z[4] = gb; z[5] = gb - ga; }
    if constexpr (PH == 1) {
        z[0] += 0.5f * gb - ga; z[1] += ga + 0.25f * gb; z[2] += 4.25f * ga - 2.5f * gb; z[3] -= 4.25f * ga + 1.25f * gb;
        z[4] += 2.f * gb - ga; z[5] += ga + gb;
    }
    if constexpr (PH == 2) {
        z[0] += 2.f * gb - 0.5f * ga; z[1] += 0.25f * ga + 4.f * gb; z[2] += 2.5f * (ga - gb); z[3] -= 1.25f * ga + 5.f * gb;
        z[4] += 0.5f * gb - 2.f * ga; z[5] += ga + gb;
    }
    if constexpr (PH == 3) {
        z[0] -= 2.f * ga + gb; z[1] += 4.f * ga; z[2] += 2.5f * ga + 5.25f * gb; z[3] -= 5.f * ga; z[4] -= 0.5f * ga + 5.25f * gb; z[5] += ga;
    }
}

// the 36-bit tile masks travel as two 32-bit halves (compile-time bit positions: no 64-bit shifts)
struct Bits36 { unsigned lo, hi; };
__device__ __forceinline__ Bits36 load_bits36(const void* table, size_t idx) {
    const uint2 v = reinterpret_cast<const uint2*>(table)[idx];
    return Bits36{v.x, v.y};
}
__device__ __forceinline__ void store_bits36(void* table, size_t idx, Bits36 b) {
    reinterpret_cast<uint2*>(table)[idx] = make_uint2(b.lo, b.hi);
}
template <int B>
__device__ __forceinline__ bool bit36(const Bits36& m) { return B < 32 ? ((m.lo >> B) & 1u) : ((m.hi >> (B - 32)) & 1u); }
template <int B>
__device__ __forceinline__ void set36(Bits36& m, bool on) {
    if constexpr (B < 32) m.lo |= (on ? 1u : 0u) << B; else m.hi |= (on ? 1u : 0u) << (B - 32);
}

// compile-time loops over the 36 pixels of a block (the bit index must be a constant expression)
template <int I, int J, typename F>
__device__ __forceinline__ void for36_step(F&& f) {
    f(std::integral_constant<int, I>{}, std::integral_constant<int, J>{});
    if constexpr (J < 5) for36_step<I, J + 1>(f);
    else if constexpr (I < 5) for36_step<I + 1, 0>(f);
}
template <typename F>
__device__ __forceinline__ void for36(F&& f) { for36_step<0, 0>(f); }
template <int R, typename F>
__device__ __forceinline__ void for6_step(F&& f) {
    f(std::integral_constant<int, R>{});
    if constexpr (R < 5) for6_step<R + 1>(f);
}
template <typename F>
__device__ __forceinline__ void for6(F&& f) { for6_step<0>(f); }
// row R of a block . its mask bits
template <int R>
__device__ __forceinline__ void mask_row6(const Bits36& m, float* y) {
    y[0] = bit36<6 * R + 0>(m) ? y[0] : 0.f; y[1] = bit36<6 * R + 1>(m) ? y[1] : 0.f; y[2] = bit36<6 * R + 2>(m) ? y[2] : 0.f;
    y[3] = bit36<6 * R + 3>(m) ? y[3] : 0.f; y[4] = bit36<6 * R + 4>(m) ? y[4] : 0.f; y[5] = bit36<6 * R + 5>(m) ? y[5] : 0.f;
}

// the tile's own 6 columns of one row: float4 + float2 (order by the parity of tx) -> own[6]
__device__ __forceinline__ void own6(bool odd, const float4& A, const float2& B, float* own) {
    own[0] = odd ? B.x : A.x; own[1] = odd ? B.y : A.y; own[2] = odd ? A.x : A.z;
    own[3] = odd ? A.y : A.w; own[4] = odd ? A.z : B.x; own[5] = odd ? A.w : B.y;
}
// (values, not a pointer: a select between y[0..3] and y[2..5] read through a pointer becomes a dynamically indexed load and the
// whole 6x6 block goes to scratch)
__device__ __forceinline__ void split6(bool odd, float y0, float y1, float y2, float y3, float y4, float y5, float4& A, float2& B) {
    A.x = odd ? y2 : y0; A.y = odd ? y3 : y1; A.z = odd ? y4 : y2; A.w = odd ? y5 : y3;
    B.x = odd ? y0 : y4; B.y = odd ? y1 : y5;
}

// PRE: the maps are PRE-activations -- the transform reads relu(x + bias[c]) (the bias + ReLU epilogue of the producing 1x1 convolution
// folded into this load) and writes the tile's 36-bit activation mask for the adjoint transform of the backward (wino6_in_t).
// H2: V is written as f16x2 split rows of V * 2^e (winograd.h), e from the bound *amax_in of the (activated) input: |V| <= 15 * 15 * bound
template <bool VEC, bool PRE, bool H2>
__device__ __forceinline__ void wino6_in_body(const WinoArgs& a, int l, float* lds) {
    const int H = a.H[l], W = a.W[l], TH = a.TH[l], TW = a.TW[l];
    const long long units = (long long)a.N * TH * TW, padded = (units + kTilePad - 1) & ~(long long)(kTilePad - 1);
    const long long t0 = (long long)(blockIdx.x - a.blk_off[l]) * 256;
    const long long u = t0 + threadIdx.x;
    const bool on = u < units;
    const long long uu = on ? u : units - 1;
    const int c = blockIdx.y;
    const size_t plane = (size_t)a.T;
    int tx, ty, n;
    tile_coords(uu, TW, TH, tx, ty, n);
    const float* p = a.maps_in[l] + ((size_t)n * a.C + c) * H * W;
    const int y0 = 6 * ty - 1, x0 = 6 * tx;   // x0 = the tile's own first column = window column 1
    const int lane = threadIdx.x & 63;
    float prb = 0.f, prs = 1.f;   // PRE: relu(x * prs + prb); prs = 1 for the per-channel bias form (fma(x, 1, b) = x + b exactly)
    if constexpr (PRE) {
        if (a.pre_affine) { const float2 sa = reinterpret_cast<const float2*>(a.pre_affine)[((size_t)l * a.N + n) * a.C + c]; prs = sa.x; prb = sa.y; }
        else prb = a.bias[c];
    }
    float h2s = 1.f;              // H2: the power-of-two scale; with PRE it rides on the affine (relu(z) * s = relu(z * s), s > 0), else on the loaded values
    if constexpr (H2) {
        const int e = h2_exponent(*a.amax_in, 2 * kH2LgBt);
        h2s = h2_pow2(e);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.scale_out[0] = h2_pow2(-e);
        if constexpr (PRE) { prs *= h2s; prb *= h2s; }
    }
    float d[8][8];
    if constexpr (VEC) {
        // every load of the window is issued before anything consumes one (one exposed HBM latency per workgroup, not one per row)
        const bool odd = tx & 1;
        const int xa = x0 + (odd ? 2 : 0), xb = x0 + (odd ? 0 : 4);
        const bool va = xa + 4 <= W, vb = xb + 2 <= W;   // W % 4 == 0: a vector is fully inside or fully outside the row
        const bool needL = lane == 0 && tx != 0, needR = (lane == 63 || u + 1 >= units) && tx != TW - 1;
        float4 A[8]; float2 B[8]; float hl[8], hr[8];
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int y = y0 + i;
            const bool yok = y >= 0 && y < H;
            const size_t ro = (size_t)(yok ? y : 0) * W;
            A[i] = (yok && va) ? *reinterpret_cast<const float4*>(p + ro + xa) : make_float4(0.f, 0.f, 0.f, 0.f);
            B[i] = (yok && vb) ? *reinterpret_cast<const float2*>(p + ro + xb) : make_float2(0.f, 0.f);
        }
        if (needL) {
            #pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int y = y0 + i;
                const bool yok = y >= 0 && y < H;
                hl[i] = yok ? p[(size_t)(yok ? y : 0) * W + x0 - 1] : 0.f;
            }
        }
        if (needR) {   // tx != TW - 1: column x0 + 6 <= 6 (TW - 1) < W
            #pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int y = y0 + i;
                const bool yok = y >= 0 && y < H;
                hr[i] = yok ? p[(size_t)(yok ? y : 0) * W + x0 + 6] : 0.f;
            }
        }
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int y = y0 + i;
            const bool yok = y >= 0 && y < H;
            float own[6];
            own6(odd, A[i], B[i], own);
            // PRE: pixels beyond the map stay the zero padding of the ACTIVATION: relu(0 + -inf) = 0, no branch
            const float pbv = PRE ? (yok ? prb : -INFINITY) : 0.f;
            if constexpr (PRE) {
                const float pa = va ? pbv : -INFINITY, pb = vb ? pbv : -INFINITY;
                #pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const bool inA = odd ? j >= 2 : j < 4;
                    own[j] = fmaxf(fmaf(own[j], prs, inA ? pa : pb), 0.f);
                }
            }
            float e0 = wave_shr1(own[5]), e7 = wave_shl1(own[0]);
            if (needL) { e0 = hl[i]; if constexpr (PRE) e0 = fmaxf(fmaf(e0, prs, pbv), 0.f); }
            if (needR) { e7 = hr[i]; if constexpr (PRE) e7 = fmaxf(fmaf(e7, prs, pbv), 0.f); }
            if (tx == 0) e0 = 0.f;
            if (tx == TW - 1) e7 = 0.f;
            d[i][0] = e0; d[i][7] = e7;
            #pragma unroll
            for (int j = 0; j < 6; ++j) d[i][j + 1] = own[j];
        }
    } else {
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int y = y0 + i;
            const bool yok = y >= 0 && y < H;
            const size_t ro = (size_t)(yok ? y : 0) * W;
            #pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int x = x0 - 1 + j;
                const bool ok = yok && x >= 0 && x < W;
                d[i][j] = ok ? p[ro + (ok ? x : 0)] : 0.f;
            }
        }
        if constexpr (PRE) {
            #pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int y = y0 + i;
                const bool yok = y >= 0 && y < H;
                #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int x = x0 - 1 + j;
                    const bool ok = yok && x >= 0 && x < W;
                    d[i][j] = fmaxf(fmaf(d[i][j], prs, ok ? prb : -INFINITY), 0.f);
                }
            }
        }
    }
    if constexpr (H2 && !PRE) {
        #pragma unroll
        for (int i = 0; i < 8; ++i)
            #pragma unroll
            for (int j = 0; j < 8; ++j) d[i][j] *= h2s;
    }
    const size_t base = (size_t)c * a.cs + (size_t)a.tile_off[l] + t0;
    const long long tend = padded - t0;  // tiles of this workgroup that exist (incl. zero pad tiles), relative to t0
    if constexpr (PRE) {
        if (a.bits_out && on) {   // the tile's own 6x6 block = window rows / columns 1..6 (pixels beyond the map are 0 -> bit 0)
            Bits36 bits{0u, 0u};
            for36([&](auto I, auto J) { set36<6 * I.value + J.value>(bits, d[I.value + 1][J.value + 1] > 0.f); });
            store_bits36(a.bits_out, (size_t)c * plane + (size_t)a.tile_off[l] + u, bits);
        }
    }
    #pragma unroll
    for (int j = 0; j < 8; ++j) {  // columns: B^T d (in place)
        const float col[8] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[6][j], d[7][j]};
        float w[8];
        bt8(col, w);
        #pragma unroll
        for (int i = 0; i < 8; ++i) d[i][j] = w[i];
    }
    #pragma unroll
    for (int ph = 0; ph < 4; ++ph) {   // rows: (B^T d) B, two frequency rows (16 planes, 16 KB) per phase
        if (ph) __syncthreads();
        #pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            float w[8];
            bt8(d[2 * ph + ii], w);
            #pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (H2) reinterpret_cast<uint32_t*>(lds)[(8 * ii + j) * 256 + threadIdx.x] = on ? h2_pack(w[j]) : 0u;
                else lds[(8 * ii + j) * 256 + threadIdx.x] = on ? w[j] : 0.f;
            }
        }
        __syncthreads();
        if constexpr (H2) stage_store_h2<16>(reinterpret_cast<const uint32_t*>(lds), reinterpret_cast<uint32_t*>(a.buf_out) + base, plane, 16 * ph, tend,
                                             a.tile_off[l] + t0);
        else stage_store<16>(lds, a.buf_out + base, plane, 16 * ph, tend);
    }
}

template <bool PRE, bool H2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LGD_W6_IN_WAVES, 8))) void wino6_in_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 256];
    const int l = wino_level(a);
    if (a.pair[l]) wino6_in_body<true, PRE, H2>(a, l, lds);
    else wino6_in_body<false, PRE, H2>(a, l, lds);
}

// two frequency rows (16 planes x 256 tiles) of M / dV as 1 KB runs: issued into registers (slab_issue) and parked in LDS later
// (slab_park), so that the loads of phase p + 1 are in flight while phase p is consumed -- at 3-4 waves per SIMD (the 6x6 kernels'
// register budget) a workgroup that waits for its slab, consumes it and only then asks for the next one leaves the memory system idle
// for a latency per phase; `m` = the slab's first tile in plane 0 of the channel
struct Slab16 { wino_vf4 q[4]; };
__device__ __forceinline__ Slab16 slab_issue(const float* m, size_t plane, int ph, long long valid) {
    Slab16 s;
    #pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = k * 256 + threadIdx.x, f = idx >> 6, q4 = idx & 63;
        s.q[k].x = s.q[k].y = s.q[k].z = s.q[k].w = 0.f;
        if (q4 * 4 < valid) s.q[k] = __builtin_nontemporal_load(reinterpret_cast<const wino_vf4*>(m + (size_t)(16 * ph + f) * plane + q4 * 4));
    }
    return s;
}
__device__ __forceinline__ void slab_park(const Slab16& s, float* lds) {
    #pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = k * 256 + threadIdx.x, f = idx >> 6, q4 = idx & 63;
        *reinterpret_cast<float4*>(&lds[f * 256 + q4 * 4]) = make_float4(s.q[k].x, s.q[k].y, s.q[k].z, s.q[k].w);
    }
}

// rows of 6 outputs at the even column x0: float4 + float2 (order by the parity of tx), each fully inside or outside the row
__device__ __forceinline__ void store_row6(float* row, int x0, int W, bool odd, float y0, float y1, float y2, float y3, float y4, float y5) {
    float4 A; float2 B;
    split6(odd, y0, y1, y2, y3, y4, y5, A, B);
    const int xa = x0 + (odd ? 2 : 0), xb = x0 + (odd ? 0 : 4);
    if (xa + 4 <= W) *reinterpret_cast<float4*>(row + xa) = A;
    if (xb + 2 <= W) *reinterpret_cast<float2*>(row + xb) = B;
}

// y = A^T m A + bias [ReLU]; writes the tile's 36-bit mask of y > 0 when bits_out is given
template <bool VEC>
__device__ __forceinline__ void wino6_out_body(const WinoArgs& a, int l, float* lds) {
    const int H = a.H[l], W = a.W[l], TH = a.TH[l], TW = a.TW[l];
    const long long units = (long long)a.N * TH * TW, padded = (units + kTilePad - 1) & ~(long long)(kTilePad - 1);
    const long long t0 = (long long)(blockIdx.x - a.blk_off[l]) * 256;
    const long long u = t0 + threadIdx.x;
    const int c = blockIdx.y;
    const size_t plane = (size_t)a.T;
    const float* m = a.buf_in + (size_t)c * a.cs + (size_t)a.tile_off[l] + t0;
    // columns first, accumulated as the frequency rows arrive (two per LDS phase): r[i][b] = sum_a A^T[i][a] M[a][b] -- 48 accumulators
    // instead of holding all 64 values and then 48 more (118 -> 5 waves per SIMD worth of registers)
    float r[6][8];
    Slab16 cur = slab_issue(m, plane, 0, padded - t0);
    #pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        if (ph) __syncthreads();   // the previous phase's LDS reads are done
        slab_park(cur, lds);
        if (ph < 3) cur = slab_issue(m, plane, ph + 1, padded - t0);   // in flight while this phase is consumed
        __syncthreads();
        #pragma unroll
        for (int b = 0; b < 8; ++b) {
            const float ga = lds[b * 256 + threadIdx.x], gb = lds[(8 + b) * 256 + threadIdx.x];   // frequency rows 2 ph, 2 ph + 1
            if (ph == 0) {          // a = 0: [1 0 0 0 0 0];  a = 1: [1 1 1 1 1 1]
                r[0][b] = ga + gb; r[1][b] = gb; r[2][b] = gb; r[3][b] = gb; r[4][b] = gb; r[5][b] = gb;
            } else if (ph == 1) {   // a = 2: [1 -1 1 -1 1 -1];  a = 3: [1 2 4 8 16 32]
                r[0][b] += ga + gb; r[1][b] += 2.f * gb - ga; r[2][b] += ga + 4.f * gb; r[3][b] += 8.f * gb - ga;
                r[4][b] += ga + 16.f * gb; r[5][b] += 32.f * gb - ga;
            } else if (ph == 2) {   // a = 4: [1 -2 4 -8 16 -32];  a = 5: [1 1/2 1/4 1/8 1/16 1/32]
                r[0][b] += ga + gb; r[1][b] += 0.5f * gb - 2.f * ga; r[2][b] += 4.f * ga + 0.25f * gb; r[3][b] += 0.125f * gb - 8.f * ga;
                r[4][b] += 16.f * ga + 0.0625f * gb; r[5][b] += 0.03125f * gb - 32.f * ga;
            } else {                // a = 6: [1 -1/2 1/4 -1/8 1/16 -1/32];  a = 7: [0 0 0 0 0 1]
                r[0][b] += ga; r[1][b] -= 0.5f * ga; r[2][b] += 0.25f * ga; r[3][b] -= 0.125f * ga; r[4][b] += 0.0625f * ga;
                r[5][b] += gb - 0.03125f * ga;
            }
        }
    }
    float am = 0.f;   // max |y| over the pixels this thread stores (a.amax_out: the next convolution's f16 scale comes from it)
    if (u < units) {
        int tx, ty, n;
        tile_coords(u, TW, TH, tx, ty, n);
        const float b = a.bias ? a.bias[c] : 0.f;
        float* p = a.maps_out[l] + ((size_t)n * a.C + c) * H * W;
        const int oy = 6 * ty, ox = 6 * tx;
        const bool odd = tx & 1;
        const bool relu = a.relu != 0;
        const bool want_max = a.amax_out != nullptr;
        Bits36 bits{0u, 0u};
        for6([&](auto R) {   // row by row: transform, bias / ReLU, mask bits, store
            constexpr int i = R.value;
            float y[6];
            at8(r[i], y);
            #pragma unroll
            for (int j = 0; j < 6; ++j) {
                y[j] += b;
                if (relu) y[j] = fmaxf(y[j], 0.f);
            }
            set36<6 * i + 0>(bits, y[0] > 0.f); set36<6 * i + 1>(bits, y[1] > 0.f); set36<6 * i + 2>(bits, y[2] > 0.f);
            set36<6 * i + 3>(bits, y[3] > 0.f); set36<6 * i + 4>(bits, y[4] > 0.f); set36<6 * i + 5>(bits, y[5] > 0.f);
            if (oy + i < H) {
                float* row = p + (size_t)(oy + i) * W;
                if constexpr (VEC) {
                    store_row6(row, ox, W, odd, y[0], y[1], y[2], y[3], y[4], y[5]);
                } else {
                    #pragma unroll
                    for (int j = 0; j < 6; ++j) if (ox + j < W) row[ox + j] = y[j];
                }
                if (want_max) {
                    #pragma unroll
                    for (int j = 0; j < 6; ++j) am = fmaxf(am, ox + j < W ? fabsf(y[j]) : 0.f);
                }
            }
        });
        if (a.bits_out) store_bits36(a.bits_out, (size_t)c * plane + (size_t)a.tile_off[l] + u, bits);
    }
    if (a.amax_out) {
        am = wave_max(am);
        __syncthreads();   // (the last phase's LDS reads are done)
        block_max_bits(a.amax_out, am, lds);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LGD_W6_OUT_WAVES, 8))) void wino6_out_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 256];
    const int l = wino_level(a);
    if (a.pair[l]) wino6_out_body<true>(a, l, lds);
    else wino6_out_body<false>(a, l, lds);
}

// g (6x6, masked) -> dM = A g A^T staged two frequency rows at a time.  H2: written as f16x2 split rows; frequency (i, j) carries the scale
// 2^(e0 - lgA(i) - lgA(j)) -- |dM[i][j]| <= rowsum_i(A) rowsum_j(A) max|g| -- with 2^e0 already on g (h2_block_scale) and the per-index powers of
// two applied here, after each pass
__device__ __forceinline__ float h2_block_scale(const WinoArgs& a) {
    const int e = h2_exponent(*a.amax_in, 0);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) a.scale_out[threadIdx.x] = h2_pow2(-e + h2_lg_a(threadIdx.x >> 3) + h2_lg_a(threadIdx.x & 7));
    return h2_pow2(e);
}
template <bool H2>
__device__ __forceinline__ void expand_block6(const float (&g)[6][6], bool on, float* lds, float* dst, size_t plane, long long tend, bool sync_first,
                                              long long t0) {
    constexpr float kA[8] = {1.f, 0.125f, 0.125f, 0.015625f, 0.015625f, 0.5f, 0.5f, 1.f};   // 2^-lgA
    float r[8][6];
    #pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {g[0][j], g[1][j], g[2][j], g[3][j], g[4][j], g[5][j]};
        float w[8];
        a8(col, w);
        #pragma unroll
        for (int i = 0; i < 8; ++i) r[i][j] = H2 ? w[i] * kA[i] : w[i];
    }
    #pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        if (ph || sync_first) __syncthreads();
        #pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            float w[8];
            a8(r[2 * ph + ii], w);
            #pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (H2) reinterpret_cast<uint32_t*>(lds)[(8 * ii + j) * 256 + threadIdx.x] = on ? h2_pack(w[j] * kA[j]) : 0u;
                else lds[(8 * ii + j) * 256 + threadIdx.x] = on ? w[j] : 0.f;
            }
        }
        __syncthreads();
        if constexpr (H2) stage_store_h2<16>(reinterpret_cast<const uint32_t*>(lds), reinterpret_cast<uint32_t*>(dst), plane, 16 * ph, tend, t0);
        else stage_store<16>(lds, dst, plane, 16 * ph, tend);
    }
}

// dM = A (dy . mask) A^T: the ONE transform of dy the backward pass needs (dU[f] = dM[f] V[f]^T, dV[f] = U[f]^T dM[f])
template <bool VEC, bool GN, bool H2>
__device__ __forceinline__ void wino6_out_t_body(const WinoArgs& a, int l, float* lds) {
    const int H = a.H[l], W = a.W[l], TH = a.TH[l], TW = a.TW[l];
    const long long units = (long long)a.N * TH * TW, padded = (units + kTilePad - 1) & ~(long long)(kTilePad - 1);
    const long long t0 = (long long)(blockIdx.x - a.blk_off[l]) * 256;
    const long long u = t0 + threadIdx.x;
    const bool on = u < units;
    const long long uu = on ? u : units - 1;
    const int c = blockIdx.y;
    const size_t plane = (size_t)a.T;
    int tx, ty, n;
    tile_coords(uu, TW, TH, tx, ty, n);
    const float* p = a.maps_in[l] + ((size_t)n * a.C + c) * H * W;
    Bits36 mb{0xffffffffu, 0xfu};
    if (a.bits_in) mb = load_bits36(a.bits_in, (size_t)c * plane + (size_t)a.tile_off[l] + uu);
    const int oy = 6 * ty, ox = 6 * tx;
    float g[6][6];
    if constexpr (GN) {
        // the maps are gradients w.r.t. the OUTPUT of the GroupNorm that follows this convolution: its backward apply
        // dy = rstd * (gamma * g - m1 - xhat * m2) = ca * g - cm - (y - mu) * cb runs here, on the block as it is loaded (the
        // gradient of the convolution output is never written or re-read); pixels outside the map stay zero
        const float* py = a.maps_in2[l] + ((size_t)n * a.C + c) * H * W;
        const float4 k = *reinterpret_cast<const float4*>(a.gn_coef + 4 * (((size_t)l * a.N + n) * a.C + c));
        auto gn = [&](float gv, float yv, bool ok) -> float { return ok ? fmaf(k.x, gv, -k.y) - (yv - k.z) * k.w : 0.f; };
        if constexpr (VEC) {
            const bool odd = tx & 1;
            const int xa = ox + (odd ? 2 : 0), xb = ox + (odd ? 0 : 4);
            const bool va = xa + 4 <= W, vb = xb + 2 <= W;
            float4 A[6], Ay[6]; float2 B[6], By[6];
            #pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool yok = oy + i < H;
                const size_t ro = (size_t)(yok ? oy + i : 0) * W;
                A[i] = (yok && va) ? ldg_stream4(p + ro + xa) : make_float4(0.f, 0.f, 0.f, 0.f);
                B[i] = (yok && vb) ? *reinterpret_cast<const float2*>(p + ro + xb) : make_float2(0.f, 0.f);
                Ay[i] = (yok && va) ? ldg_stream4(py + ro + xa) : make_float4(0.f, 0.f, 0.f, 0.f);
                By[i] = (yok && vb) ? *reinterpret_cast<const float2*>(py + ro + xb) : make_float2(0.f, 0.f);
            }
            #pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool yok = oy + i < H, oa = yok && va, ob = yok && vb;
                A[i] = make_float4(gn(A[i].x, Ay[i].x, oa), gn(A[i].y, Ay[i].y, oa), gn(A[i].z, Ay[i].z, oa), gn(A[i].w, Ay[i].w, oa));
                B[i] = make_float2(gn(B[i].x, By[i].x, ob), gn(B[i].y, By[i].y, ob));
                own6(odd, A[i], B[i], g[i]);
            }
        } else {
            #pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool yok = oy + i < H;
                const size_t ro = (size_t)(yok ? oy + i : 0) * W;
                #pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const bool ok = yok && ox + j < W;
                    const size_t o = ro + (ok ? ox + j : 0);
                    g[i][j] = gn(ok ? p[o] : 0.f, ok ? py[o] : 0.f, ok);
                }
            }
        }
    } else if constexpr (VEC) {
        const bool odd = tx & 1;
        const int xa = ox + (odd ? 2 : 0), xb = ox + (odd ? 0 : 4);
        const bool va = xa + 4 <= W, vb = xb + 2 <= W;
        float4 A[6]; float2 B[6];
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bool yok = oy + i < H;
            const size_t ro = (size_t)(yok ? oy + i : 0) * W;
            A[i] = (yok && va) ? ldg_stream4(p + ro + xa) : make_float4(0.f, 0.f, 0.f, 0.f);
            B[i] = (yok && vb) ? *reinterpret_cast<const float2*>(p + ro + xb) : make_float2(0.f, 0.f);
        }
        #pragma unroll
        for (int i = 0; i < 6; ++i) own6(odd, A[i], B[i], g[i]);
    } else {
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bool yok = oy + i < H;
            const size_t ro = (size_t)(yok ? oy + i : 0) * W;
            #pragma unroll
            for (int j = 0; j < 6; ++j) {
                const bool ok = yok && ox + j < W;
                g[i][j] = ok ? p[ro + (ok ? ox + j : 0)] : 0.f;
            }
        }
    }
    if constexpr (H2) {
        const float s0 = h2_block_scale(a);
        for36([&](auto I, auto J) { g[I.value][J.value] = bit36<6 * I.value + J.value>(mb) ? g[I.value][J.value] * s0 : 0.f; });
    } else {
        for36([&](auto I, auto J) { g[I.value][J.value] = bit36<6 * I.value + J.value>(mb) ? g[I.value][J.value] : 0.f; });
    }
    expand_block6<H2>(g, on, lds, a.buf_out + (size_t)c * a.cs + (size_t)a.tile_off[l] + t0, plane, padded - t0, false, a.tile_off[l] + t0);
}

template <bool H2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LGD_W6_OUTT_WAVES, 8))) void wino6_out_t_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 256];
    const int l = wino_level(a);
    if (a.pair[l]) wino6_out_t_body<true, false, H2>(a, l, lds);
    else wino6_out_t_body<false, false, H2>(a, l, lds);
}

// the same with the backward apply of a GroupNorm that follows the convolution folded into the load (WinoArgs::gn_coef)
template <bool H2>
__global__ __launch_bounds__(256) void wino6_out_t_gn_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 256];
    const int l = wino_level(a);
    if (a.pair[l]) wino6_out_t_body<true, true, H2>(a, l, lds);
    else wino6_out_t_body<false, true, H2>(a, l, lds);
}

// dx = adjoint of wino6_in: the 8x8 windows Z_t = B G_t B^T (G = dV) of neighbouring tiles overlap by two pixels and are summed
// where they do.  Written as a GATHER, one thread per tile producing its own 6x6 block of dx, so nothing is exchanged or accumulated
// in memory: window row / column 0 of a tile depends only on frequency row / column 0 (column 0 of B^T is e_0) and window row /
// column 7 only on frequency row / column 7 (column 7 of B^T is e_7), so the block needs, besides the tile's own 64 values, 8 values
// of each edge neighbour and 1 of each corner neighbour -- from the LDS slab when the neighbour tile is inside the workgroup's
// 256-tile run, else from loads issued BEFORE the barrier together with the slab.
// Order of the two passes: a frequency ROW a (8 values, delivered two rows per LDS phase) is first taken through the horizontal
// transform -- h = (B g_a)[1..6], plus the left tile's G[a][7] in column 0 and the right tile's G[a][0] in column 5 -- and then
// accumulated into the block with the vertical coefficients of row a: z[r][.] += B^T[a][r+1] h.  36 accumulators instead of the 60
// of the column-first order (t[6][8] + the two neighbour columns), and fewer operations (8 horizontal transforms instead of 6 + 96 FMAs).
template <int A>
__device__ __forceinline__ void vacc8(const float (&h)[6], float (&z)[6][6]) {
    constexpr float C[8][6] = {{0.f, -5.25f, 0.f, 5.25f, 0.f, -1.f},      {1.f, 1.f, -4.25f, -4.25f, 1.f, 1.f},
                               {-1.f, 1.f, 4.25f, -4.25f, -1.f, 1.f},     {0.5f, 0.25f, -2.5f, -1.25f, 2.f, 1.f},
                               {-0.5f, 0.25f, 2.5f, -1.25f, -2.f, 1.f},   {2.f, 4.f, -2.5f, -5.f, 0.5f, 1.f},
                               {-2.f, 4.f, 2.5f, -5.f, -0.5f, 1.f},       {-1.f, 0.f, 5.25f, 0.f, -5.25f, 0.f}};   // B^T[a][1..6]
    #pragma unroll
    for (int r = 0; r < 6; ++r) {
        if (C[A][r] == 0.f) continue;
        #pragma unroll
        for (int j = 0; j < 6; ++j) z[r][j] = fmaf(C[A][r], h[j], z[r][j]);
    }
}

// what one phase (frequency rows 2 PH, 2 PH + 1) needs from memory: the workgroup's slab and, on the edge waves, the values of
// neighbour tiles OUTSIDE the slab (the first / last TW + 1 threads' vertical neighbours, thread 0's left and thread 255's right one)
struct InTLoads { wino_vf4 q[4]; float eL[2], eR[2], eV[8], eVl, eVr; };

template <int PH>
__device__ __forceinline__ InTLoads wino6_in_t_issue(const float* m, size_t plane, int nvalid, int TW, bool hasL, bool hasR, bool hasU, bool hasD) {
    const int tid = threadIdx.x;
    const float* mp = m + (size_t)(16 * PH) * plane;   // wave-uniform base; everything below is a 32-bit offset from it
    const int ip = (int)plane;                          // 16 planes of one channel: < 2^31 elements for any map that fits the HBM
    InTLoads ld;
    #pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = k * 256 + tid, f = idx >> 6, q4 = idx & 63;
        ld.q[k].x = ld.q[k].y = ld.q[k].z = ld.q[k].w = 0.f;
        if (q4 * 4 < nvalid) ld.q[k] = __builtin_nontemporal_load(reinterpret_cast<const wino_vf4*>(mp + (f * ip + q4 * 4)));
    }
    // branch-free per lane, only the edge WAVES issue these loads (wave-uniform test); a lane of such a wave whose neighbour is inside
    // the slab (or does not exist) re-reads its own tile
    const int wb = tid & ~63, own = min(tid, nvalid - 1);
    auto far = [&](int fl, int d, bool need) -> float {
        const int li = tid + d;
        return mp[fl * ip + ((need && (li < 0 || li >= 256)) ? li : own)];
    };
    ld.eL[0] = ld.eL[1] = ld.eR[0] = ld.eR[1] = ld.eVl = ld.eVr = 0.f;
    #pragma unroll
    for (int b = 0; b < 8; ++b) ld.eV[b] = 0.f;
    if (wb == 0) { ld.eL[0] = far(7, -1, hasL); ld.eL[1] = far(15, -1, hasL); }
    if (wb == 192) { ld.eR[0] = far(0, 1, hasR); ld.eR[1] = far(8, 1, hasR); }
    if constexpr (PH == 0 || PH == 3) {
        constexpr int vrow = PH == 3 ? 8 : 0;              // frequency row 7 (upper neighbour) lives in planes 8..15 of phase 3
        const int vd = PH == 3 ? -TW : TW;                 // phase 0: frequency row 0 of the LOWER tile row; phase 3: row 7 of the UPPER one
        const bool hasV = PH == 3 ? hasU : hasD;
        if (PH == 3 ? (wb - TW - 1 < 0) : (wb + 63 + TW + 1 >= 256)) {
            #pragma unroll
            for (int b = 0; b < 8; ++b) ld.eV[b] = far(vrow + b, vd, hasV);
            ld.eVl = far(vrow + 7, vd - 1, hasV && hasL);
            ld.eVr = far(vrow, vd + 1, hasV && hasR);
        }
    }
    return ld;
}

__device__ __forceinline__ void wino6_in_t_park(const InTLoads& ld, float* lds) {
    #pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = k * 256 + threadIdx.x, f = idx >> 6, q4 = idx & 63;
        *reinterpret_cast<float4*>(&lds[f * 256 + q4 * 4]) = make_float4(ld.q[k].x, ld.q[k].y, ld.q[k].z, ld.q[k].w);
    }
}

template <int PH>
__device__ __forceinline__ void wino6_in_t_consume(const InTLoads& ld, const float* lds, int TW, bool hasL, bool hasR, bool hasU, bool hasD,
                                                   float (&z)[6][6]) {
    const int tid = threadIdx.x;
    // value of local plane fl (global plane 16 PH + fl) of the tile d positions further along the level's tile run
    auto pick = [&](int fl, int d, bool need, float e) -> float {
        const int li = tid + d;
        const float v = lds[fl * 256 + min(max(li, 0), 255)];
        return !need ? 0.f : ((li >= 0 && li < 256) ? v : e);
    };
    {   // frequency row 2 PH
        float g[8], h[6];
        #pragma unroll
        for (int b = 0; b < 8; ++b) g[b] = lds[b * 256 + tid];
        b8mid(g, h);
        h[0] += pick(7, -1, hasL, ld.eL[0]);    // the left tile's window column 7 = its frequency column 7
        h[5] += pick(0, 1, hasR, ld.eR[0]);     // the right tile's window column 0 = its frequency column 0
        vacc8<2 * PH>(h, z);
    }
    {   // frequency row 2 PH + 1
        float g[8], h[6];
        #pragma unroll
        for (int b = 0; b < 8; ++b) g[b] = lds[(8 + b) * 256 + tid];
        b8mid(g, h);
        h[0] += pick(15, -1, hasL, ld.eL[1]);
        h[5] += pick(8, 1, hasR, ld.eR[1]);
        vacc8<2 * PH + 1>(h, z);
    }
    if constexpr (PH == 0 || PH == 3) {
        // PH 0: the lower tile's window row 0 (= its frequency row 0, B^T[0][0] = 1) is this block's row 5;
        // PH 3: the upper tile's window row 7 (= its frequency row 7, B^T[7][7] = 1) is this block's row 0
        constexpr int vrow = PH == 3 ? 8 : 0;
        const int vd = PH == 3 ? -TW : TW;
        const bool hasV = PH == 3 ? hasU : hasD;
        float g[8], h[6];
        #pragma unroll
        for (int b = 0; b < 8; ++b) g[b] = pick(vrow + b, vd, hasV, ld.eV[b]);
        b8mid(g, h);
        h[0] += pick(vrow + 7, vd - 1, hasV && hasL, ld.eVl);
        h[5] += pick(vrow, vd + 1, hasV && hasR, ld.eVr);
        #pragma unroll
        for (int j = 0; j < 6; ++j) z[PH == 0 ? 5 : 0][j] += h[j];
    }
}

// FUSE: instead of storing the block, apply the producing convolution's ReLU mask and transform it straight into dM = A (dx . mask) A^T
// of THAT convolution -- the backward link between two convolutions of a conv -> ReLU -> conv chain
template <bool VEC, bool FUSE, bool H2>
__device__ __forceinline__ void wino6_in_t_body(const WinoArgs& a, int l, float* lds) {
    const int H = a.H[l], W = a.W[l], TH = a.TH[l], TW = a.TW[l];
    const long long units = (long long)a.N * TH * TW, padded = (units + kTilePad - 1) & ~(long long)(kTilePad - 1);
    const long long t0 = (long long)(blockIdx.x - a.blk_off[l]) * 256;
    const long long u = t0 + threadIdx.x;
    const bool on = u < units;
    const long long uu = on ? u : units - 1;
    const int c = blockIdx.y;
    const size_t plane = (size_t)a.T;
    int tx, ty, n;
    tile_coords(uu, TW, TH, tx, ty, n);
    const bool hasL = on && tx > 0, hasR = on && tx < TW - 1, hasU = on && ty > 0, hasD = on && ty < TH - 1;
    const float* m = a.buf_in + (size_t)c * a.cs + (size_t)a.tile_off[l] + t0;
    float z[6][6];   // the tile's own block of dx
    #pragma unroll
    for (int r = 0; r < 6; ++r)
        #pragma unroll
        for (int j = 0; j < 6; ++j) z[r][j] = 0.f;
    const int nvalid = (int)min(padded - t0, 256LL);
    // software pipeline over the four LDS phases: the loads of phase p + 1 are issued before phase p is consumed
    InTLoads la = wino6_in_t_issue<0>(m, plane, nvalid, TW, hasL, hasR, hasU, hasD);
    wino6_in_t_park(la, lds);
    InTLoads lb = wino6_in_t_issue<1>(m, plane, nvalid, TW, hasL, hasR, hasU, hasD);
    __syncthreads();
    wino6_in_t_consume<0>(la, lds, TW, hasL, hasR, hasU, hasD, z);
    __syncthreads();
    wino6_in_t_park(lb, lds);
    la = wino6_in_t_issue<2>(m, plane, nvalid, TW, hasL, hasR, hasU, hasD);
    __syncthreads();
    wino6_in_t_consume<1>(lb, lds, TW, hasL, hasR, hasU, hasD, z);
    __syncthreads();
    wino6_in_t_park(la, lds);
    lb = wino6_in_t_issue<3>(m, plane, nvalid, TW, hasL, hasR, hasU, hasD);
    __syncthreads();
    wino6_in_t_consume<2>(la, lds, TW, hasL, hasR, hasU, hasD, z);
    __syncthreads();
    wino6_in_t_park(lb, lds);
    __syncthreads();
    wino6_in_t_consume<3>(lb, lds, TW, hasL, hasR, hasU, hasD, z);
    const int oy = 6 * ty, ox = 6 * tx;
    // optional mask: the activation bits the PRE input transform wrote (the maps were pre-activations: dx is the gradient of the RAW
    // map) or, FUSE, the producing convolution's ReLU bits
    Bits36 mb{0xffffffffu, 0xfu};
    if (a.bits_in) mb = load_bits36(a.bits_in, (size_t)c * plane + (size_t)a.tile_off[l] + uu);
    for36([&](auto I, auto J) { z[I.value][J.value] = bit36<6 * I.value + J.value>(mb) ? z[I.value][J.value] : 0.f; });
    if constexpr (!FUSE) {
        if (a.amax_out) {   // max |dx| over the map (pixels of overhanging tiles beyond it excluded), for the consumer's f16 scale
            float am = 0.f;
            #pragma unroll
            for (int r = 0; r < 6; ++r)
                #pragma unroll
                for (int j = 0; j < 6; ++j) am = fmaxf(am, (on && oy + r < H && ox + j < W) ? fabsf(z[r][j]) : 0.f);
            am = wave_max(am);
            __syncthreads();   // (the last gather phase's LDS reads are done)
            block_max_bits(a.amax_out, am, lds);
        }
        if (!on) return;
        float* p = a.maps_out[l] + ((size_t)n * a.C + c) * H * W;
        const bool odd = tx & 1;
        #pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (oy + r >= H) continue;
            float* row = p + (size_t)(oy + r) * W;
            if constexpr (VEC) {
                store_row6(row, ox, W, odd, z[r][0], z[r][1], z[r][2], z[r][3], z[r][4], z[r][5]);
            } else {
                #pragma unroll
                for (int j = 0; j < 6; ++j) if (ox + j < W) row[ox + j] = z[r][j];
            }
        }
    } else {
        // tiles may overhang the map: the gradient of pixels beyond it is not part of dx
        #pragma unroll
        for (int r = 0; r < 6; ++r)
            #pragma unroll
            for (int j = 0; j < 6; ++j) z[r][j] = (oy + r < H && ox + j < W) ? z[r][j] : 0.f;
        if constexpr (H2) {   // *amax_in bounds |dx| (lgd_h2_link_bound from the per-frequency maxima of dV)
            const float s0 = h2_block_scale(a);
            #pragma unroll
            for (int r = 0; r < 6; ++r)
                #pragma unroll
                for (int j = 0; j < 6; ++j) z[r][j] *= s0;
        }
        // (the slab is still being read by the last gather phase: expand_block6 syncs before its first LDS write)
        expand_block6<H2>(z, on, lds, a.buf_out + (size_t)c * a.cs + (size_t)a.tile_off[l] + t0, plane, padded - t0, true, a.tile_off[l] + t0);
    }
}

template <bool FUSE, bool H2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LGD_W6_INT_WAVES, 8))) void wino6_in_t_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 256];
    const int l = wino_level(a);
    if (a.pair[l]) wino6_in_t_body<true, FUSE, H2>(a, l, lds);
    else wino6_in_t_body<false, FUSE, H2>(a, l, lds);
}

// ------------------------------------------------------------------------------------------------------------------
// Filter transforms of F(6x6,3x3): U = G (s . g) G^T for every (C_out, C_in) pair, written twice -- U [64][Co][Ci] for the forward
// product and U^T [64][Ci][Co] for dV = U^T dM -- and the adjoint dg = s . G^T dU G (s: optional frozen per-output-channel scale).
__device__ __forceinline__ void g8(float a, float b, float c, float* o) {   // G [a b c]^T
    const float s = (a + c) * (-2.f / 9.f), t = b * (2.f / 9.f);
    const float u = a * (1.f / 90.f) + c * (2.f / 45.f), v = b * (1.f / 45.f);
    const float p = a * (32.f / 45.f) + c * (8.f / 45.f), q = b * (16.f / 45.f);
    o[0] = a; o[1] = s - t; o[2] = s + t; o[3] = u + v; o[4] = u - v; o[5] = p + q; o[6] = p - q; o[7] = c;
}
__device__ __forceinline__ void g8t(const float* m, float* o) {            // G^T m, m[8] -> o[3]
    const float s12 = m[1] + m[2], d21 = m[2] - m[1], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
    o[0] = m[0] - s12 * (2.f / 9.f) + s34 * (1.f / 90.f) + s56 * (32.f / 45.f);
    o[1] = d21 * (2.f / 9.f) + d34 * (1.f / 45.f) + d56 * (16.f / 45.f);
    o[2] = m[7] - s12 * (2.f / 9.f) + s34 * (2.f / 45.f) + s56 * (8.f / 45.f);
}

// 16 x 16 (co, ci) pairs per workgroup; U rows are written straight (ci fastest), U^T through an LDS tile (co fastest), 32 frequencies at a time
__global__ __launch_bounds__(256) void wino6_filter_fwd_kernel(FilterArgs a) {
    __shared__ float tile[32][16][17];
    const int cl = threadIdx.x & 15, ol = threadIdx.x >> 4;
    const int ci = blockIdx.x * 16 + cl, co = blockIdx.y * 16 + ol;
    const bool on = ci < a.Ci && co < a.Co;
    float u[8][8];
    {
        float g[9];
        const float sc = (on && a.scale) ? a.scale[co] : 1.f;
        #pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = on ? a.w[((size_t)co * a.Ci + ci) * 9 + i] * sc : 0.f;
        float r[8][3];
        #pragma unroll
        for (int j = 0; j < 3; ++j) {   // columns: G g
            float o[8];
            g8(g[j], g[3 + j], g[6 + j], o);
            #pragma unroll
            for (int i = 0; i < 8; ++i) r[i][j] = o[i];
        }
        #pragma unroll
        for (int i = 0; i < 8; ++i) g8(r[i][0], r[i][1], r[i][2], u[i]);   // rows: (G g) G^T
    }
    const int co2 = blockIdx.y * 16 + cl, ci2 = blockIdx.x * 16 + ol;   // transposed roles: co fastest
    #pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half && a.Ut) __syncthreads();
        #pragma unroll
        for (int f = 0; f < 32; ++f) {
            const float v = u[(32 * half + f) / 8][(32 * half + f) % 8];
            if (on) a.U[(size_t)(32 * half + f) * a.u_plane + (size_t)co * a.Ci + ci] = v;
            if (a.Ut) tile[f][ol][cl] = v;
        }
        if (!a.Ut) continue;   // U alone (wave-uniform): the caller's dV GEMM takes U^T as a transposed operand
        __syncthreads();
        if (co2 < a.Co && ci2 < a.Ci) {
            #pragma unroll
            for (int f = 0; f < 32; ++f) a.Ut[(size_t)(32 * half + f) * a.ut_plane + (size_t)ci2 * a.ut_ld + co2] = tile[f][cl][ol];
        }
    }
}

// The same transform written as the bf16x3 operand IMAGES of csrc/gemm3.hip instead of fp32 U: image of U (rows co, k = ci) for the forward
// product M = U V, image of U^T (rows ci, k = co) for dV = U^T dM -- the filter is split where it is produced, not by a pass that
// re-reads U (58 launches of ~20 us per step at BASELINE config 2).  Co % 16 == Ci % 16 == row0 % 16 == 0 (host-checked): every
// workgroup owns whole 16-deep k-steps of both images, nothing is guarded.  32 frequencies at a time through the LDS tile.
// H2: the f16x2 images of csrc/h2.hip -- [batch][k-step of 16][2 pieces][32-row block][1 KB] of U[f] * 2^e(f), e(f) from the bound
// max|w . scale| * rowsum_i(|G|) rowsum_j(|G|) (*a.amax_in; row sums 1, 2/3, 2/3, 7/90, 7/90, 56/45, 56/45, 1 <= 2^(0, 0, 0, -3, -3, 1, 1, 0));
// workgroup (0, 0) records the 64 inverse scales in a.inv_out
__host__ __device__ constexpr int h2_lg_g(int i) { return i == 3 || i == 4 ? -3 : (i == 5 || i == 6 ? 1 : 0); }
__device__ __forceinline__ long h2_image_off(long b, int ktp, int rbp, int m, int k) {   // byte offset of the h fragment holding (m, k)
    return ((((b * ktp + (k >> 4)) * 2) * rbp + (m >> 5)) << 10) + ((((k >> 3) & 1) * 32 + (m & 31)) << 4);
}
__device__ __forceinline__ void store_split8_h2(const float* x, float s, char* d, long rbp) {
    uint32_t h[4], m[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t p0 = h2_pack(x[2 * e] * s), p1 = h2_pack(x[2 * e + 1] * s);
        h[e] = __builtin_amdgcn_perm(p1, p0, 0x05040100u);
        m[e] = __builtin_amdgcn_perm(p1, p0, 0x07060302u);
    }
    *reinterpret_cast<lgd_u32x4*>(d) = (lgd_u32x4){h[0], h[1], h[2], h[3]};
    *reinterpret_cast<lgd_u32x4*>(d + rbp * 1024) = (lgd_u32x4){m[0], m[1], m[2], m[3]};
}

template <bool H2>
__global__ __launch_bounds__(256) void wino6_filter_img_kernel(FilterArgs a) {
    __shared__ float tile[32][16][17];
    const int cl = threadIdx.x & 15, ol = threadIdx.x >> 4;
    const int ci = blockIdx.x * 16 + cl, co = blockIdx.y * 16 + ol;
    float u[8][8];
    {
        float g[9];
        const float sc = a.scale ? a.scale[co] : 1.f;
        #pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = a.w[((size_t)co * a.Ci + ci) * 9 + i] * sc;
        float r[8][3];
        #pragma unroll
        for (int j = 0; j < 3; ++j) {
            float o[8];
            g8(g[j], g[3 + j], g[6 + j], o);
            #pragma unroll
            for (int i = 0; i < 8; ++i) r[i][j] = o[i];
        }
        #pragma unroll
        for (int i = 0; i < 8; ++i) g8(r[i][0], r[i][1], r[i][2], u[i]);
    }
    int e0 = 0;
    if constexpr (H2) {
        e0 = h2_exponent(*a.amax_in, 0);
        if (a.inv_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64)
            a.inv_out[threadIdx.x] = h2_pow2(-(e0 - h2_lg_g(threadIdx.x >> 3) - h2_lg_g(threadIdx.x & 7)));
    }
    const int ktp_f = a.Ci >> 4, rbp_f = (a.Ct + 31) >> 5;     // image of U:   M = Ct, K = Ci
    const int ktp_b = a.Ct >> 4, rbp_b = (a.Ci + 31) >> 5;     // image of U^T: M = Ci, K = Ct
    const int co0 = a.row0 + blockIdx.y * 16, ci0 = blockIdx.x * 16;
    #pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        #pragma unroll
        for (int f = 0; f < 32; ++f) tile[f][ol][cl] = u[(32 * half + f) / 8][(32 * half + f) % 8];
        __syncthreads();
        #pragma unroll
        for (int j = 0; j < 4; ++j) {   // 32 frequencies x 32 fragments (16 rows x 2 k-groups) per image: 4 per thread
            const int q = threadIdx.x + 256 * j, fl = q >> 5, x = q & 15, g = (q >> 4) & 1;
            const long f = 32 * half + fl;
            float s = 1.f;
            if constexpr (H2) s = h2_pow2(e0 - h2_lg_g((int)f >> 3) - h2_lg_g((int)f & 7));
            float v[8];
            if (a.img_fwd) {
                #pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tile[fl][x][g * 8 + e];
                if constexpr (H2) store_split8_h2(v, s, a.img_fwd + h2_image_off(f, ktp_f, rbp_f, co0 + x, ci0 + g * 8), rbp_f);
                else store_split8(v, a.img_fwd + gemm3_image_off(f, ktp_f, rbp_f, co0 + x, ci0 + g * 8), rbp_f);
            }
            if (a.img_bwd) {
                #pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tile[fl][g * 8 + e][x];
                if constexpr (H2) store_split8_h2(v, s, a.img_bwd + h2_image_off(f, ktp_b, rbp_b, ci0 + x, co0 + g * 8), rbp_b);
                else store_split8(v, a.img_bwd + gemm3_image_off(f, ktp_b, rbp_b, ci0 + x, co0 + g * 8), rbp_b);
            }
        }
    }
}

__global__ __launch_bounds__(256) void wino6_filter_bwd_kernel(FilterArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.Co * a.Ci) return;
    const int co = (int)(idx / a.Ci);
    float r[3][8];
    #pragma unroll
    for (int b = 0; b < 8; ++b) {   // columns: G^T dU
        float col[8];
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* q = a.dU + (size_t)(8 * i + b) * a.u_plane + idx;
            float v = q[0];
            for (int s = 1; s < a.S; ++s) v += q[(size_t)s * a.part_stride];   // split-K partials of csrc/h2.hip, fixed order
            col[i] = v;
        }
        float o[3];
        g8t(col, o);
        r[0][b] = o[0]; r[1][b] = o[1]; r[2][b] = o[2];
    }
    const float sc = a.scale ? a.scale[co] : 1.f;
    #pragma unroll
    for (int i = 0; i < 3; ++i) {
        float o[3];
        g8t(r[i], o);                // rows: (G^T dU) G
        #pragma unroll
        for (int j = 0; j < 3; ++j) a.dw[idx * 9 + 3 * i + j] = o[j] * sc;
    }
}

void wino6_launch_in(const WinoArgs& a, unsigned blocks, bool pre, hipStream_t st) {
    const dim3 grid(blocks, a.C), block(256);
    if (a.h2) {
        if (pre) { LGD_LAUNCH("wino_in_kernel", (wino6_in_kernel<true, true>), grid, block, 0, st, a); }
        else { LGD_LAUNCH("wino_in_kernel", (wino6_in_kernel<false, true>), grid, block, 0, st, a); }
    } else if (pre) { LGD_LAUNCH("wino_in_kernel", (wino6_in_kernel<true, false>), grid, block, 0, st, a); }
    else { LGD_LAUNCH("wino_in_kernel", (wino6_in_kernel<false, false>), grid, block, 0, st, a); }
}
void wino6_launch_out(const WinoArgs& a, unsigned blocks, hipStream_t st) {
    LGD_LAUNCH("wino_out_kernel", wino6_out_kernel, dim3(blocks, a.C), dim3(256), 0, st, a);
}
void wino6_launch_out_t(const WinoArgs& a, unsigned blocks, hipStream_t st) {
    if (a.gn_coef && a.h2) { LGD_LAUNCH("wino_out_t_gn_kernel", wino6_out_t_gn_kernel<true>, dim3(blocks, a.C), dim3(256), 0, st, a); }
    else if (a.gn_coef) { LGD_LAUNCH("wino_out_t_gn_kernel", wino6_out_t_gn_kernel<false>, dim3(blocks, a.C), dim3(256), 0, st, a); }
    else if (a.h2) { LGD_LAUNCH("wino_out_t_kernel", wino6_out_t_kernel<true>, dim3(blocks, a.C), dim3(256), 0, st, a); }
    else { LGD_LAUNCH("wino_out_t_kernel", wino6_out_t_kernel<false>, dim3(blocks, a.C), dim3(256), 0, st, a); }
}
void wino6_launch_in_t(const WinoArgs& a, unsigned blocks, bool fuse, hipStream_t st) {
    const dim3 grid(blocks, a.C), block(256);
    if (fuse && a.h2) { LGD_LAUNCH("wino_in_t_out_t_kernel", (wino6_in_t_kernel<true, true>), grid, block, 0, st, a); }
    else if (fuse) { LGD_LAUNCH("wino_in_t_out_t_kernel", (wino6_in_t_kernel<true, false>), grid, block, 0, st, a); }
    else { LGD_LAUNCH("wino_in_t_kernel", (wino6_in_t_kernel<false, false>), grid, block, 0, st, a); }
}
void wino6_launch_filter_fwd(const FilterArgs& a, hipStream_t st) {
    LGD_LAUNCH("wino_filter_kernel", wino6_filter_fwd_kernel, dim3((a.Ci + 15) / 16, (a.Co + 15) / 16), dim3(256), 0, st, a);
}
void wino6_launch_filter_img(const FilterArgs& a, hipStream_t st) {
    if (a.amax_in) { LGD_LAUNCH("wino_filter_img_kernel", wino6_filter_img_kernel<true>, dim3(a.Ci / 16, a.Co / 16), dim3(256), 0, st, a); }
    else { LGD_LAUNCH("wino_filter_img_kernel", wino6_filter_img_kernel<false>, dim3(a.Ci / 16, a.Co / 16), dim3(256), 0, st, a); }
}
void wino6_launch_filter_bwd(const FilterArgs& a, hipStream_t st) {
    const long long n = (long long)a.Co * a.Ci;
    LGD_LAUNCH("wino_filter_bwd_kernel", wino6_filter_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}

}  // namespace lgd
