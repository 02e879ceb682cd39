// Winograd data transforms for the 3x3 / stride 1 / pad 1 convolutions of the LGD path
//   [ref: dynamic_teacher.py:57,61,67-73 (student_proj_2D, local_inst_proj_2D, refinement_module),
//    models/adapters/sequential_convs.py:10-12, and the student head re-run on the teacher features,
//    distillator.py:107-109 -> retinanet.py:36-43].
// 87 % of the first end-to-end training step was fp32 3x3 convolutions on the library's kernels (100-116 TFLOP/s effective).  The
// minimal-filtering form F(m x m, 3x3) needs (m+2)^2 / m^2 multiplies per output pixel instead of 9, and its (m+2)^2 independent
// (C_out x C_in) x (C_in x tiles) products are plain library GEMMs (rocBLAS / hipBLASLt fp32 MFMA).  What is left is pure HBM
// streaming, which is what these kernels do (this file: m = 4, 36 frequencies; winograd6.hip: m = 6, 64 frequencies):
//   wino_in     : x_l (N,C,H_l,W_l)        -> V [C][nf][T]       V = B^T d B   per (m+2)^2 input window (stride m, halo 1)
//   wino_out    : M [C][nf][T], bias       -> y_l (N,C,H_l,W_l)  Y = A^T m A   per tile (m x m outputs) [+bias] [ReLU] [+ mask bits]
//   wino_out_t  : dy_l [. mask bits]       -> dM [C][nf][T]      dM = A dy A^T (adjoint of wino_out: the one expansion of dy)
//   wino_in_t   : dV [C][nf][T]            -> dx_l               adjoint of wino_in, written as a gather per m x m block
//   wino_in_t_out_t: dV -> dM of the producing convolution (the backward link of a conv -> ReLU -> conv chain, no map in between)
// Every module on the path applies ONE filter to all pyramid levels, and a Winograd tile does not care which level it
// came from: the tiles of all L levels are concatenated along T (every level padded with zero tiles to a multiple of 16), so one
// conv over the pyramid is one transform launch + one batched GEMM + one transform launch, and p6/p7 (1.6 % of the pixels, but a
// third of the launches on the per-level library path) ride along for free.  Tile index is the fastest axis everywhere:
// all global accesses are coalesced along x and nothing is transposed.  The frequency planes of one channel are adjacent
// ([C][nf][T]: the GEMM of frequency f sees a (C x T) matrix with row stride nf*T) -- a workgroup's nf output chunks then
// lie within one 45 KB..1.6 MB neighbourhood instead of nf planes 88 MB apart (measured 4.6 -> 5.2 TB/s on the p3 input
// transform, tools/lab/wino4_lab.hip).
#include "winograd.h"

namespace lgd {

// ------------------------------------------------------------------------------------------------------------------
// F(4x4, 3x3): 6x6 input windows at stride 4, 36 frequencies, 4x4 outputs per tile (points 0, +-1, +-2, inf).
// 4x (instead of 2.25x) fewer multiplies than the direct form and a 2.25x (instead of 4x) expansion of the
// activations into V / M: the GEMMs shrink to 0.56x and the transform traffic to ~0.65x of F(2x2,3x3).  fp32 rounding
// grows to ~1e-5 of the output scale per convolution (F(2x2): ~6e-7, direct: ~3e-7); through the path's GroupNorm /
// InstanceNorm stages the teacher features stay within 3e-6 of the reference golden (tolerance 1e-4).
// One thread per tile: an aligned float4 per window row (W % 4 == 0) + the two halo columns from the neighbour lanes.
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void bt6(const float* d, float* t) {
    const float a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = a + b; t[2] = a - b; t[3] = c + e; t[4] = c - e;
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// A^T m: 6 -> 4
__device__ __forceinline__ void at6(const float* m, float* y) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34; y[1] = d12 + 2.f * d34; y[2] = s12 + 4.f * s34; y[3] = d12 + 8.f * d34 + m[5];
}
// A g: 4 -> 6 (adjoint of at6)
__device__ __forceinline__ void a6(const float* g, float* r) {
    const float e = g[0] + g[2], o = g[1] + g[3], e4 = g[0] + 4.f * g[2], o2 = 2.f * g[1] + 8.f * g[3];
    r[0] = g[0]; r[1] = e + o; r[2] = e - o; r[3] = e4 + o2; r[4] = e4 - o2; r[5] = g[3];
}

// The nf = 36 values of the workgroup's 256 tiles go through LDS two frequency rows (12 planes, 12 KB) at a time so that every
// frequency plane is written / read as ONE 1 KB run (float4 per lane) instead of 256 B per wave, while 3x more workgroups stay
// resident than with a 36-plane slab (kTilePad, stage_store: winograd.h).
//
// PRE: the maps are PRE-activations -- the transform reads relu(x + bias[c]) (the bias + ReLU epilogue of the producing 1x1 convolution
// folded into this load: conv1 -> FrozenBN -> ReLU -> conv2 of a bottleneck block) and writes the tile's 16-bit
// activation mask (bit 4*i+j = its own 4x4 block's pixel (i, j) > 0) for the adjoint transform of the backward (wino4_in_t).
template <bool VEC, bool PRE>
__device__ __forceinline__ void wino4_in_body(const WinoArgs& a, int l, float* lds) {
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
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    const int lane = threadIdx.x & 63;
    float prb = 0.f, prs = 1.f;   // PRE: relu(x * prs + prb); prs = 1 for the per-channel bias form (fma(x, 1, b) = x + b exactly)
    if constexpr (PRE) {
        if (a.pre_affine) { const float2 sa = reinterpret_cast<const float2*>(a.pre_affine)[((size_t)l * a.N + n) * a.C + c]; prs = sa.x; prb = sa.y; }
        else prb = a.bias[c];
    }
    float d[6][6];
    if constexpr (VEC) {
        // Phase 1: EVERY load of the 6x6 window is issued before anything consumes one -- six aligned float4 rows and, on the wave's end
        // lanes, the two halo columns.  Written as one loop (load, activate, DPP halo exchange per row) the compiler put a wait behind
        // each row's load: 6 (plain) to 18 (PRE) exposed HBM latencies per workgroup instead of one.
        float4 m[6];
        float hl[6], hr[6];
        const bool needL = lane == 0 && tx != 0, needR = (lane == 63 || u + 1 >= units) && tx != TW - 1;
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int y = y0 + i;
            const bool yok = y >= 0 && y < H;
            const size_t ro = (size_t)(yok ? y : 0) * W;
            m[i] = yok ? *reinterpret_cast<const float4*>(p + ro + x0 + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (needL) {
            #pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int y = y0 + i;
                const bool yok = y >= 0 && y < H;
                hl[i] = yok ? p[(size_t)(yok ? y : 0) * W + x0] : 0.f;
            }
        }
        if (needR) {
            #pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int y = y0 + i;
                const bool yok = y >= 0 && y < H;
                hr[i] = yok ? p[(size_t)(yok ? y : 0) * W + x0 + 5] : 0.f;
            }
        }
        // Phase 2: folded activation, halo exchange, window assembly
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int y = y0 + i;
            const bool yok = y >= 0 && y < H;
            float4 v = m[i];
            // PRE: rows beyond the map stay the zero padding of the ACTIVATION: relu(0 + -inf) = 0, no branch
            const float pbv = PRE ? (yok ? prb : -INFINITY) : 0.f;
            if constexpr (PRE) { v.x = fmaxf(fmaf(v.x, prs, pbv), 0.f); v.y = fmaxf(fmaf(v.y, prs, pbv), 0.f); v.z = fmaxf(fmaf(v.z, prs, pbv), 0.f); v.w = fmaxf(fmaf(v.w, prs, pbv), 0.f); }
            // halo columns = the neighbour lanes' edge values (same image row unless first / last tile of the row): one DPP move each
            // (measured equal to ds_bpermute shuffles, 136.1 vs 136.4 us)
            float e0 = wave_shr1(v.w), e5 = wave_shl1(v.x);
            if (needL) { e0 = hl[i]; if constexpr (PRE) e0 = fmaxf(fmaf(e0, prs, pbv), 0.f); }
            if (needR) { e5 = hr[i]; if constexpr (PRE) e5 = fmaxf(fmaf(e5, prs, pbv), 0.f); }
            if (tx == 0) e0 = 0.f;
            if (tx == TW - 1) e5 = 0.f;
            d[i][0] = e0; d[i][1] = v.x; d[i][2] = v.y; d[i][3] = v.z; d[i][4] = v.w; d[i][5] = e5;
        }
    } else {
        // W % 4 != 0 (res5 / p5 / p7 at 800x1344): 36 dword loads, all issued before the first is consumed (same reason as above)
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int y = y0 + i;
            const bool yok = y >= 0 && y < H;
            const size_t ro = (size_t)(yok ? y : 0) * W;
            #pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int x = x0 + j;
                const bool ok = yok && x >= 0 && x < W;
                d[i][j] = ok ? p[ro + x] : 0.f;
            }
        }
        if constexpr (PRE) {
            #pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int y = y0 + i;
                const bool yok = y >= 0 && y < H;
                #pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int x = x0 + j;
                    const bool ok = yok && x >= 0 && x < W;
                    d[i][j] = fmaxf(fmaf(d[i][j], prs, ok ? prb : -INFINITY), 0.f);
                }
            }
        }
    }
    const size_t base = (size_t)c * a.cs + (size_t)a.tile_off[l] + t0;
    const long long tend = padded - t0;  // tiles of this workgroup that exist (incl. zero pad tiles), relative to t0
    if constexpr (PRE) {
        if (a.bits_out && on) {   // the tile's own 4x4 block = window rows / columns 1..4 (pixels beyond the map are 0 -> bit 0)
            unsigned bits = 0u;
            #pragma unroll
            for (int i = 0; i < 4; ++i)
                #pragma unroll
                for (int j = 0; j < 4; ++j) bits |= (d[i + 1][j + 1] > 0.f ? 1u : 0u) << (4 * i + j);
            static_cast<unsigned short*>(a.bits_out)[(size_t)c * plane + (size_t)a.tile_off[l] + u] = (unsigned short)bits;
        }
    }
    float r[6][6];
    #pragma unroll
    for (int j = 0; j < 6; ++j) {  // columns: B^T d
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float w[6];
        bt6(col, w);
        #pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = w[i];
    }
    #pragma unroll
    for (int ph = 0; ph < 3; ++ph) {   // rows: (B^T d) B, two frequency rows per phase
        if (ph) __syncthreads();
        #pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            float w[6];
            bt6(r[2 * ph + ii], w);
            #pragma unroll
            for (int j = 0; j < 6; ++j) lds[(6 * ii + j) * 256 + threadIdx.x] = on ? w[j] : 0.f;
        }
        __syncthreads();
        stage_store<12>(lds, a.buf_out + base, plane, 12 * ph, tend);
    }
}

template <bool PRE>
__global__ __launch_bounds__(256) void wino4_in_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[12 * 256];
    const int l = wino_level(a);
    if (a.pair[l]) wino4_in_body<true, PRE>(a, l, lds);
    else wino4_in_body<false, PRE>(a, l, lds);
}

// y = A^T m A + bias [ReLU]; reads of M staged through LDS two frequency rows (12 KB) at a time: 1 KB runs instead of 256 B per wave,
// measured 83 -> 77 us in the step (HBM-cold 109 -> 100 us = 6.0 TB/s); a 36-plane slab (36 KB, a third of the resident workgroups)
// was slower than direct 256 B loads (124 us)
template <bool VEC>
__device__ __forceinline__ void wino4_out_body(const WinoArgs& a, int l, float* lds) {
    const int H = a.H[l], W = a.W[l], TH = a.TH[l], TW = a.TW[l];
    const long long units = (long long)a.N * TH * TW, padded = (units + kTilePad - 1) & ~(long long)(kTilePad - 1);
    const long long t0 = (long long)(blockIdx.x - a.blk_off[l]) * 256;
    const long long u = t0 + threadIdx.x;
    const int c = blockIdx.y;
    const size_t plane = (size_t)a.T;
    const float* m = a.buf_in + (size_t)c * a.cs + (size_t)a.tile_off[l] + t0;
    float mm[6][6];
    #pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
        if (ph) __syncthreads();
        #pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = k * 256 + threadIdx.x, f = idx >> 6, q4 = idx & 63;
            wino_vf4 q; q.x = q.y = q.z = q.w = 0.f;
            if (t0 + q4 * 4 < padded)
                q = __builtin_nontemporal_load(reinterpret_cast<const wino_vf4*>(m + (size_t)(12 * ph + f) * plane + q4 * 4));
            *reinterpret_cast<float4*>(&lds[f * 256 + q4 * 4]) = make_float4(q.x, q.y, q.z, q.w);
        }
        __syncthreads();
        #pragma unroll
        for (int ii = 0; ii < 2; ++ii)
            #pragma unroll
            for (int j = 0; j < 6; ++j) mm[2 * ph + ii][j] = lds[(6 * ii + j) * 256 + threadIdx.x];
    }
    if (u >= units) return;
    int tx, ty, n;
    tile_coords(u, TW, TH, tx, ty, n);
    float r[4][6];
    #pragma unroll
    for (int j = 0; j < 6; ++j) {  // columns: A^T m
        const float col[6] = {mm[0][j], mm[1][j], mm[2][j], mm[3][j], mm[4][j], mm[5][j]};
        float w[4];
        at6(col, w);
        r[0][j] = w[0]; r[1][j] = w[1]; r[2][j] = w[2]; r[3][j] = w[3];
    }
    const float b = a.bias ? a.bias[c] : 0.f;
    float* p = a.maps_out[l] + ((size_t)n * a.C + c) * H * W;
    const int oy = 4 * ty, ox = 4 * tx;
    unsigned bits = 0u;
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        float y[4];
        at6(r[i], y);
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] += b;
            if (a.relu) y[j] = fmaxf(y[j], 0.f);
            bits |= (y[j] > 0.f ? 1u : 0u) << (4 * i + j);
        }
        if (i == 3 && a.bits_out) static_cast<unsigned short*>(a.bits_out)[(size_t)c * plane + (size_t)a.tile_off[l] + u] = (unsigned short)bits;
        if (oy + i >= H) continue;
        float* row = p + (size_t)(oy + i) * W + ox;
        if constexpr (VEC) {
            *reinterpret_cast<float4*>(row) = make_float4(y[0], y[1], y[2], y[3]);
        } else {
            #pragma unroll
            for (int j = 0; j < 4; ++j) if (ox + j < W) row[j] = y[j];
        }
    }
}

__global__ __launch_bounds__(256) void wino4_out_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[12 * 256];
    const int l = wino_level(a);
    if (a.pair[l]) wino4_out_body<true>(a, l, lds);
    else wino4_out_body<false>(a, l, lds);
}

// dM = A dy A^T alone: the ONE transform of dy the backward pass needs.  Both backward products hang off it --
// dU[f] = dM[f] V[f]^T (weight gradient) and dV[f] = U[f]^T dM[f] (input gradient in the frequency domain, brought back by
// wino4_in_t below) -- so dy is expanded once (2.25x) instead of twice (the rotated-filter form needs B^T dy B as well).
// A tile's 4x4 block is four aligned float4 rows (W % 4 == 0); staged two frequency rows (12 KB) at a time like wino4_in.
template <bool VEC>
__device__ __forceinline__ void wino4_out_t_body(const WinoArgs& a, int l, float* lds) {
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
    const unsigned mb = a.bits_in ? static_cast<const unsigned short*>(a.bits_in)[(size_t)c * plane + (size_t)a.tile_off[l] + uu] : 0xffffu;
    // all loads of the 4x4 block first, then the mask (one exposed HBM latency per workgroup)
    float g[4][4];
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = 4 * ty + i;
        const bool yok = y < H;
        const size_t ro = (size_t)(yok ? y : 0) * W + 4 * tx;
        if constexpr (VEC) {
            const float4 m = yok ? ldg_stream4(p + ro) : make_float4(0.f, 0.f, 0.f, 0.f);
            g[i][0] = m.x; g[i][1] = m.y; g[i][2] = m.z; g[i][3] = m.w;
        } else {
            #pragma unroll
            for (int j = 0; j < 4; ++j) g[i][j] = (yok && 4 * tx + j < W) ? p[ro + j] : 0.f;
        }
    }
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned nib = mb >> (4 * i);
        #pragma unroll
        for (int j = 0; j < 4; ++j) g[i][j] = ((nib >> j) & 1u) ? g[i][j] : 0.f;
    }
    float r[6][4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float col[4] = {g[0][j], g[1][j], g[2][j], g[3][j]};
        float w[6];
        a6(col, w);
        #pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = w[i];
    }
    float* dst = a.buf_out + (size_t)c * a.cs + (size_t)a.tile_off[l] + t0;
    #pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
        if (ph) __syncthreads();
        #pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            float w[6];
            a6(r[2 * ph + ii], w);
            #pragma unroll
            for (int j = 0; j < 6; ++j) lds[(6 * ii + j) * 256 + threadIdx.x] = on ? w[j] : 0.f;
        }
        __syncthreads();
        stage_store<12>(lds, dst, plane, 12 * ph, padded - t0);
    }
}

__global__ __launch_bounds__(256) void wino4_out_t_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[12 * 256];
    const int l = wino_level(a);
    if (a.pair[l]) wino4_out_t_body<true>(a, l, lds);
    else wino4_out_t_body<false>(a, l, lds);
}

// dx = adjoint of wino4_in: the 6x6 windows Z_t = B G_t B^T (G = dV, B = (B^T)^T) of neighbouring tiles overlap by two
// pixels and are summed where they do.  Written as a GATHER, one thread per tile producing its own 4x4 block of dx, so
// nothing is exchanged or accumulated in memory: window row / column 0 of a tile depends only on frequency row / column 0
// (B's row 0 is [4 0 0 0 0 0]) and window row / column 5 only on frequency row / column 5 ([0 0 0 0 0 1]), so the block needs,
// besides the tile's own 36 values, 6 values of each edge neighbour and 1 of each corner neighbour (64 loads, the extra 28
// from lines this workgroup or its neighbour streams anyway).  The 36 planes are read as 1 KB runs through 12 KB of LDS like
// wino4_out; neighbour values come from the LDS slab when the neighbour tile is inside the workgroup's 256-tile run.
//   z = B g:  z0 = 4 g0, z1 = 4(g2-g1) + 2(g4-g3) + 4 g5, z2 = -5 g0 - 4(g1+g2) - (g3+g4), z3 = (g1-g2) + 2(g3-g4) - 5 g5,
//             z4 = g0+g1+g2+g3+g4, z5 = g5
__device__ __forceinline__ void b6mid(const float* g, float* z) {   // z1..z4 (the rows / columns inside the tile's own block)
    const float s12 = g[1] + g[2], d21 = g[2] - g[1], s34 = g[3] + g[4], d43 = g[4] - g[3];
    z[0] = 4.f * d21 + 2.f * d43 + 4.f * g[5];
    z[1] = -5.f * g[0] - 4.f * s12 - s34;
    z[2] = -d21 - 2.f * d43 - 5.f * g[5];
    z[3] = g[0] + s12 + s34;
}

// rows (2 PH, 2 PH + 1) of g added into z1..z4 = (B g)[1..4]: the frequency rows arrive two at a time (one LDS phase), so the
// column pass accumulates instead of holding all 36 values (half the registers of the two-pass form)
template <int PH>
__device__ __forceinline__ void b6acc(float ga, float gb, float* z) {
    if constexpr (PH == 0) { z[0] = -4.f * gb; z[1] = -5.f * ga - 4.f * gb; z[2] = gb; z[3] = ga + gb; }
    if constexpr (PH == 1) { z[0] += 4.f * ga - 2.f * gb; z[1] -= 4.f * ga + gb; z[2] += 2.f * gb - ga; z[3] += ga + gb; }
    if constexpr (PH == 2) { z[0] += 2.f * ga + 4.f * gb; z[1] -= ga; z[2] -= 2.f * ga + 5.f * gb; z[3] += ga; }
}

template <bool VEC, int PH>
__device__ __forceinline__ void wino4_in_t_phase(const float* m, size_t plane, int nvalid, float* lds,
                                                 int TW, bool hasL, bool hasR, bool hasU, bool hasD,
                                                 float (&t)[4][6], float (&tl)[4], float (&tr)[4]) {
    const int tid = threadIdx.x;
    const float* mp = m + (size_t)(12 * PH) * plane;   // wave-uniform base; everything below is a 32-bit offset from it
    const int ip = (int)plane;                          // 12 planes of one channel: < 2^31 elements for any map that fits the HBM
    if (PH) __syncthreads();
    wino_vf4 q[3];
    #pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = k * 256 + tid, f = idx >> 6, q4 = idx & 63;
        q[k].x = q[k].y = q[k].z = q[k].w = 0.f;
        if (q4 * 4 < nvalid) q[k] = __builtin_nontemporal_load(reinterpret_cast<const wino_vf4*>(mp + (f * ip + q4 * 4)));
    }
    // Neighbour tiles outside the workgroup's 256-tile slab (the first / last TW+1 threads' vertical neighbours, thread 0's left
    // and thread 255's right one) are loaded from memory HERE, together with the slab, so that their latency is not paid after
    // the barrier.  Branch-free per lane: only the edge WAVES issue these loads (wave-uniform test); a lane of such a wave whose
    // neighbour is inside the slab (or does not exist) re-reads its own tile -- a line the slab load touches anyway.
    const int wb = tid & ~63, own = min(tid, nvalid - 1);
    auto far = [&](int fl, int d, bool need) -> float {
        const int li = tid + d;
        return mp[fl * ip + ((need && (li < 0 || li >= 256)) ? li : own)];
    };
    // value of local plane fl (global plane 12 PH + fl) of the tile d positions further along the level's tile run
    auto pick = [&](int fl, int d, bool need, float e) -> float {
        const int li = tid + d;
        const float v = lds[fl * 256 + min(max(li, 0), 255)];
        return !need ? 0.f : ((li >= 0 && li < 256) ? v : e);
    };
    float eL0 = 0.f, eL1 = 0.f, eR0 = 0.f, eR1 = 0.f;
    if (wb == 0) { eL0 = far(5, -1, hasL); eL1 = far(11, -1, hasL); }
    if (wb == 192) { eR0 = far(0, 1, hasR); eR1 = far(6, 1, hasR); }
    constexpr int vrow = PH == 2 ? 6 : 0;             // frequency row 5 (upper neighbour) lives in planes 6..11 of phase 2
    const int vd = PH == 2 ? -TW : TW;                 // phase 0: frequency row 0 of the LOWER tile row; phase 2: row 5 of the UPPER one
    const bool hasV = PH == 2 ? hasU : hasD;
    float eV[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, eVl = 0.f, eVr = 0.f;
    if constexpr (PH != 1) {
        if (PH == 2 ? (wb - TW - 1 < 0) : (wb + 63 + TW + 1 >= 256)) {
            #pragma unroll
            for (int b = 0; b < 6; ++b) eV[b] = far(vrow + b, vd, hasV);
            eVl = far(vrow + 5, vd - 1, hasV && hasL);
            eVr = far(vrow, vd + 1, hasV && hasR);
        }
    }
    #pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = k * 256 + tid, f = idx >> 6, q4 = idx & 63;
        *reinterpret_cast<float4*>(&lds[f * 256 + q4 * 4]) = make_float4(q[k].x, q[k].y, q[k].z, q[k].w);
    }
    __syncthreads();
    #pragma unroll
    for (int b = 0; b < 6; ++b) {
        float z[4] = {t[0][b], t[1][b], t[2][b], t[3][b]};
        b6acc<PH>(lds[b * 256 + tid], lds[(6 + b) * 256 + tid], z);
        t[0][b] = z[0]; t[1][b] = z[1]; t[2][b] = z[2]; t[3][b] = z[3];
    }
    b6acc<PH>(pick(5, -1, hasL, eL0), pick(11, -1, hasL, eL1), tl);   // frequency column 5 of the left tile
    b6acc<PH>(pick(0, 1, hasR, eR0), pick(6, 1, hasR, eR1), tr);      // frequency column 0 of the right tile
    if constexpr (PH == 0) {   // the lower tile's window row 0 = this block's row 3 (B[0][0] = 4)
        #pragma unroll
        for (int b = 0; b < 6; ++b) t[3][b] += 4.f * pick(b, vd, hasV, eV[b]);
        tl[3] += 4.f * pick(5, vd - 1, hasV && hasL, eVl);
        tr[3] += 4.f * pick(0, vd + 1, hasV && hasR, eVr);
    }
    if constexpr (PH == 2) {   // the upper tile's window row 5 = this block's row 0 (B[5][5] = 1)
        #pragma unroll
        for (int b = 0; b < 6; ++b) t[0][b] += pick(6 + b, vd, hasV, eV[b]);
        tl[0] += pick(11, vd - 1, hasV && hasL, eVl);
        tr[0] += pick(6, vd + 1, hasV && hasR, eVr);
    }
}

// the tile's own 4x4 block of dx = the adjoint input transform of dV (everything above); FUSE: instead of storing it, apply the
// producing convolution's ReLU mask and transform it straight into dM = A (dx . mask) A^T of THAT convolution -- the backward link
// between two convolutions of a conv -> ReLU -> conv chain whose intermediate map has no other consumer: the gradient map is
// neither written nor re-read (4.5 instead of 6.5 maps of traffic per link).
template <bool VEC, bool FUSE>
__device__ __forceinline__ void wino4_in_t_body(const WinoArgs& a, int l, float* lds) {
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
    // t[r][b]: window row r+1 (= block row r) of B G per frequency column b, incl. the vertical neighbours' rows;
    // tl / tr: the same for frequency column 5 of the left tile / column 0 of the right tile
    float t[4][6], tl[4], tr[4];
    const int nvalid = (int)min(padded - t0, 256LL);   // tiles of the slab that exist (>= 4: a workgroup starts below `units`)
    wino4_in_t_phase<VEC, 0>(m, plane, nvalid, lds, TW, hasL, hasR, hasU, hasD, t, tl, tr);
    wino4_in_t_phase<VEC, 1>(m, plane, nvalid, lds, TW, hasL, hasR, hasU, hasD, t, tl, tr);
    wino4_in_t_phase<VEC, 2>(m, plane, nvalid, lds, TW, hasL, hasR, hasU, hasD, t, tl, tr);
    const int oy = 4 * ty, ox = 4 * tx;
    if constexpr (!FUSE) {
        if (!on) return;
        float* p = a.maps_out[l] + ((size_t)n * a.C + c) * H * W;
        // optional: the activation mask the PRE input transform wrote (the maps were pre-activations: dx is the gradient of the RAW map)
        const unsigned mb = a.bits_in ? static_cast<const unsigned short*>(a.bits_in)[(size_t)c * plane + (size_t)a.tile_off[l] + u] : 0xffffu;
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y[4];
            b6mid(t[r], y);         // rows: window columns 1..4 of (B G) B^T
            y[0] += tl[r];          // the left tile's window column 5 (B[5][5] = 1)
            y[3] += 4.f * tr[r];    // the right tile's window column 0 (B[0][0] = 4)
            #pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = ((mb >> (4 * r + j)) & 1u) ? y[j] : 0.f;
            if (oy + r >= H) continue;
            float* row = p + (size_t)(oy + r) * W + ox;
            if constexpr (VEC) {
                *reinterpret_cast<float4*>(row) = make_float4(y[0], y[1], y[2], y[3]);
            } else {
                #pragma unroll
                for (int j = 0; j < 4; ++j) if (ox + j < W) row[j] = y[j];
            }
        }
    } else {
        // mask: the producing conv's ReLU bits (all ones without a ReLU) and the map's extent (tiles may overhang it)
        const unsigned mb = a.bits_in ? static_cast<const unsigned short*>(a.bits_in)[(size_t)c * plane + (size_t)a.tile_off[l] + uu] : 0xffffu;
        float g[4][4];
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y[4];
            b6mid(t[r], y);
            y[0] += tl[r];
            y[3] += 4.f * tr[r];
            #pragma unroll
            for (int j = 0; j < 4; ++j)
                g[r][j] = (((mb >> (4 * r + j)) & 1u) && oy + r < H && ox + j < W) ? y[j] : 0.f;
        }
        float rr[6][4];
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float col[4] = {g[0][j], g[1][j], g[2][j], g[3][j]};
            float w[6];
            a6(col, w);
            #pragma unroll
            for (int i = 0; i < 6; ++i) rr[i][j] = w[i];
        }
        float* dst = a.buf_out + (size_t)c * a.cs + (size_t)a.tile_off[l] + t0;
        #pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            __syncthreads();   // the slab is still being read by the last gather phase / the previous store phase
            #pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                float w[6];
                a6(rr[2 * ph + ii], w);
                #pragma unroll
                for (int j = 0; j < 6; ++j) lds[(6 * ii + j) * 256 + threadIdx.x] = on ? w[j] : 0.f;
            }
            __syncthreads();
            stage_store<12>(lds, dst, plane, 12 * ph, padded - t0);
        }
    }
}

template <bool FUSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void wino4_in_t_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[12 * 256];
    const int l = wino_level(a);
    if (a.pair[l]) wino4_in_t_body<true, FUSE>(a, l, lds);
    else wino4_in_t_body<false, FUSE>(a, l, lds);
}

// ------------------------------------------------------------------------------------------------------------------
// Filter transforms of F(4x4,3x3): U = G (s . g) G^T for every (C_out, C_in) pair, written twice -- U [36][Co][Ci] for the forward
// product and U^T [36][Ci][Co] for dV = U^T dM -- and the adjoint dg = s . G^T dU G.  s = the frozen per-output-channel scale of a
// FrozenBN that follows the convolution (NULL: none): folding it here costs nothing, where the host-side form paid a scale kernel, a
// GEMM against kron(G,G), a transposing copy and, backward, another GEMM and another scale kernel per convolution and step.
//   G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
__device__ __forceinline__ void g6(float a, float b, float c, float* o) {   // G [a b c]^T
    const float s = (a + c) * (1.f / 6.f), t = b * (1.f / 6.f), u = a * (1.f / 24.f) + c * (1.f / 6.f), v = b * (1.f / 12.f);
    o[0] = a * 0.25f; o[1] = -s - t; o[2] = t - s; o[3] = u + v; o[4] = u - v; o[5] = c;
}
__device__ __forceinline__ void g6t(const float* m, float* o) {            // G^T m, m[6] -> o[3]
    const float p = m[1] + m[2], q = m[2] - m[1], r = m[3] + m[4], d = m[3] - m[4];
    o[0] = m[0] * 0.25f - p * (1.f / 6.f) + r * (1.f / 24.f);
    o[1] = q * (1.f / 6.f) + d * (1.f / 12.f);
    o[2] = (r - p) * (1.f / 6.f) + m[5];
}

// 16 x 16 (co, ci) pairs per workgroup; U rows are written straight (ci fastest), U^T through an LDS tile (co fastest)
__global__ __launch_bounds__(256) void wino4_filter_fwd_kernel(FilterArgs a) {
    __shared__ float tile[36][16][17];
    const int cl = threadIdx.x & 15, ol = threadIdx.x >> 4;
    const int ci = blockIdx.x * 16 + cl, co = blockIdx.y * 16 + ol;
    const bool on = ci < a.Ci && co < a.Co;
    float u[6][6];
    {
        float g[9];
        const float sc = (on && a.scale) ? a.scale[co] : 1.f;
        #pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = on ? a.w[((size_t)co * a.Ci + ci) * 9 + i] * sc : 0.f;
        float r[6][3];
        #pragma unroll
        for (int j = 0; j < 3; ++j) {   // columns: G g
            float o[6];
            g6(g[j], g[3 + j], g[6 + j], o);
            #pragma unroll
            for (int i = 0; i < 6; ++i) r[i][j] = o[i];
        }
        #pragma unroll
        for (int i = 0; i < 6; ++i) g6(r[i][0], r[i][1], r[i][2], u[i]);   // rows: (G g) G^T
    }
    #pragma unroll
    for (int f = 0; f < 36; ++f) {
        const float v = u[f / 6][f % 6];
        if (on) a.U[(size_t)f * a.u_plane + (size_t)co * a.Ci + ci] = v;
        tile[f][ol][cl] = v;
    }
    if (!a.Ut) return;   // U alone: the caller's dV GEMM takes U^T as a transposed operand
    __syncthreads();
    const int co2 = blockIdx.y * 16 + cl, ci2 = blockIdx.x * 16 + ol;   // transposed roles: co fastest
    if (co2 < a.Co && ci2 < a.Ci) {
        #pragma unroll
        for (int f = 0; f < 36; ++f) a.Ut[(size_t)f * a.ut_plane + (size_t)ci2 * a.ut_ld + co2] = tile[f][cl][ol];
    }
}

__global__ __launch_bounds__(256) void wino4_filter_bwd_kernel(FilterArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.Co * a.Ci) return;
    const int co = (int)(idx / a.Ci);
    float m[6][6];
    #pragma unroll
    for (int f = 0; f < 36; ++f) m[f / 6][f % 6] = a.dU[(size_t)f * a.u_plane + idx];
    float r[3][6];
    #pragma unroll
    for (int b = 0; b < 6; ++b) {   // columns: G^T dU
        const float col[6] = {m[0][b], m[1][b], m[2][b], m[3][b], m[4][b], m[5][b]};
        float o[3];
        g6t(col, o);
        r[0][b] = o[0]; r[1][b] = o[1]; r[2][b] = o[2];
    }
    const float sc = a.scale ? a.scale[co] : 1.f;
    #pragma unroll
    for (int i = 0; i < 3; ++i) {
        float o[3];
        g6t(r[i], o);                // rows: (G^T dU) G
        #pragma unroll
        for (int j = 0; j < 3; ++j) a.dw[idx * 9 + 3 * i + j] = o[j] * sc;
    }
}

long long wino_level_tiles(int N, int H, int W, int tile) {
    return (((long long)N * ((H + tile - 1) / tile) * ((W + tile - 1) / tile)) + kTilePad - 1) & ~(long long)(kTilePad - 1);
}

// fills the per-level tables (tile = 4 or 6)
int wino_fill(WinoArgs& a, const int32_t* level_hw, int L, int N, int C, int tile, unsigned* blocks) {
    if (!level_hw || L < 1 || L > LGD_MAX_LEVELS || N < 1 || C < 1 || C > 65535 || (tile != 4 && tile != 6)) return LGD_EINVAL;
    a.L = L; a.N = N; a.C = C; a.relu = 0;
    a.bias = nullptr; a.pre_affine = nullptr; a.gn_coef = nullptr; a.buf_in = nullptr; a.buf_out = nullptr; a.bits_out = nullptr; a.bits_in = nullptr;
    a.amax_in = nullptr; a.scale_out = nullptr; a.amax_out = nullptr; a.h2 = 0;
    long long off = 0;
    unsigned blk = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.maps_in[l] = nullptr; a.maps_out[l] = nullptr; a.maps_in2[l] = nullptr;
        a.H[l] = a.W[l] = a.TH[l] = a.TW[l] = a.pair[l] = 0; a.tile_off[l] = 0; a.blk_off[l] = 0;
    }
    for (int l = 0; l < L; ++l) {
        const int H = level_hw[2 * l], W = level_hw[2 * l + 1];
        if (H < 1 || W < 1) return LGD_EINVAL;
        a.H[l] = H; a.W[l] = W; a.TH[l] = (H + tile - 1) / tile; a.TW[l] = (W + tile - 1) / tile;
        a.pair[l] = W % 4 == 0 ? 1 : 0;  // aligned vector rows (tile 4: float4; tile 6: float4 + float2)
        a.tile_off[l] = off;
        a.blk_off[l] = blk;
        const long long units = (long long)N * a.TH[l] * a.TW[l];
        if (units >= (1LL << 31) - 256) return LGD_EINVAL;   // the kernels index a level's tiles in 32 bits
        blk += (unsigned)((units + 255) / 256);
        off += wino_level_tiles(N, H, W, tile);
    }
    a.blk_off[L] = blk;
    a.T = off;
    a.cs = (long long)(tile + 2) * (tile + 2) * off;
    *blocks = blk;
    return LGD_OK;
}

}  // namespace lgd

extern "C" {

size_t lgd_wino_tiles(const int32_t* level_hw_host, int L, int N, int tile) {
    if (!level_hw_host || L < 1 || N < 1 || (tile != 4 && tile != 6)) return 0;
    long long t = 0;
    for (int l = 0; l < L; ++l) t += lgd::wino_level_tiles(N, level_hw_host[2 * l], level_hw_host[2 * l + 1], tile);
    return (size_t)t;
}

size_t lgd_wino_mask_bytes(int tile) { return tile == 6 ? 8 : (tile == 4 ? 2 : 0); }

int lgd_wino_in(const float* const* x_host, const int32_t* level_hw_host, int L, int N, int C, int tile, float* V,
                const float* pre_bias, const float* pre_affine, void* pre_bits, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!x_host || !V || lgd::wino_fill(a, level_hw_host, L, N, C, tile, &blocks) != LGD_OK) return LGD_EINVAL;
    if ((pre_bits && !pre_bias && !pre_affine) || (pre_bias && pre_affine)) return LGD_EINVAL;
    a.bias = pre_bias; a.pre_affine = pre_affine; a.bits_out = pre_bits;
    for (int l = 0; l < L; ++l) {
        if (!x_host[l]) return LGD_EINVAL;
        a.maps_in[l] = x_host[l];
    }
    a.buf_out = V;
    const dim3 grid(blocks, C), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (tile == 6) lgd::wino6_launch_in(a, blocks, pre_bias != nullptr || pre_affine != nullptr, st);
    else if (pre_bias || pre_affine) { LGD_LAUNCH("wino_in_kernel", (lgd::wino4_in_kernel<true>), grid, block, 0, st, a); }
    else { LGD_LAUNCH("wino_in_kernel", (lgd::wino4_in_kernel<false>), grid, block, 0, st, a); }
    return lgd::check_launch();
}

int lgd_wino_out(const float* M, const float* bias, const int32_t* level_hw_host, int L, int N, int C, int tile,
                 int relu, float* const* y_host, void* relu_bits, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!M || !y_host || lgd::wino_fill(a, level_hw_host, L, N, C, tile, &blocks) != LGD_OK) return LGD_EINVAL;
    a.bits_out = relu_bits;
    for (int l = 0; l < L; ++l) {
        if (!y_host[l]) return LGD_EINVAL;
        a.maps_out[l] = y_host[l];
    }
    a.buf_in = M; a.bias = bias; a.relu = relu ? 1 : 0;
    if (tile == 6) lgd::wino6_launch_out(a, blocks, (hipStream_t)stream);
    else { LGD_LAUNCH("wino_out_kernel", lgd::wino4_out_kernel, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a); }
    return lgd::check_launch();
}

int lgd_wino_out_t(const float* const* dy_host, const void* relu_bits, const int32_t* level_hw_host, int L, int N, int C, int tile,
                   float* dM, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!dy_host || !dM || lgd::wino_fill(a, level_hw_host, L, N, C, tile, &blocks) != LGD_OK) return LGD_EINVAL;
    a.bits_in = relu_bits;
    for (int l = 0; l < L; ++l) {
        if (!dy_host[l]) return LGD_EINVAL;
        a.maps_in[l] = dy_host[l];
    }
    a.buf_out = dM;
    if (tile == 6) lgd::wino6_launch_out_t(a, blocks, (hipStream_t)stream);
    else { LGD_LAUNCH("wino_out_t_kernel", lgd::wino4_out_t_kernel, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a); }
    return lgd::check_launch();
}

int lgd_wino_out_t_gn(const float* const* g_host, const float* const* y_host, const float* coef, const int32_t* level_hw_host, int L,
                      int N, int C, int tile, float* dM, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!g_host || !y_host || !coef || !dM || tile != 6 || lgd::wino_fill(a, level_hw_host, L, N, C, tile, &blocks) != LGD_OK)
        return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!g_host[l] || !y_host[l]) return LGD_EINVAL;
        a.maps_in[l] = g_host[l];
        a.maps_in2[l] = y_host[l];
    }
    a.gn_coef = coef;
    a.buf_out = dM;
    lgd::wino6_launch_out_t(a, blocks, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_wino_in_t(const float* dV, const int32_t* level_hw_host, int L, int N, int C, int tile, float* const* dx_host,
                  const void* pre_bits, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!dV || !dx_host || lgd::wino_fill(a, level_hw_host, L, N, C, tile, &blocks) != LGD_OK) return LGD_EINVAL;
    a.bits_in = pre_bits;
    for (int l = 0; l < L; ++l) {
        if (!dx_host[l]) return LGD_EINVAL;
        a.maps_out[l] = dx_host[l];
    }
    a.buf_in = dV;
    if (tile == 6) lgd::wino6_launch_in_t(a, blocks, false, (hipStream_t)stream);
    else { LGD_LAUNCH("wino_in_t_kernel", lgd::wino4_in_t_kernel<false>, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a); }
    return lgd::check_launch();
}

int lgd_wino_filter_fwd(const float* w, const float* scale, int Co, int Ci, int tile, float* U, long long u_plane, float* Ut,
                        long long ut_ld, long long ut_plane, void* stream) {
    if (!w || !U || Co < 1 || Ci < 1 || (tile != 4 && tile != 6) || u_plane < (long long)Co * Ci
        || (Ut && (ut_ld < Co || ut_plane < (long long)Ci * ut_ld))) return LGD_EINVAL;
    lgd::FilterArgs a{};
    a.w = w; a.scale = scale; a.U = U; a.Ut = Ut; a.u_plane = u_plane; a.ut_plane = ut_plane; a.ut_ld = ut_ld; a.Co = Co; a.Ci = Ci;
    if (tile == 6) lgd::wino6_launch_filter_fwd(a, (hipStream_t)stream);
    else { LGD_LAUNCH("wino_filter_kernel", lgd::wino4_filter_fwd_kernel, dim3((Ci + 15) / 16, (Co + 15) / 16), dim3(256), 0, (hipStream_t)stream, a); }
    return lgd::check_launch();
}

int lgd_wino_filter_images(const float* w, const float* scale, int Co, int Ci, int tile, int row0, int Ct, void* img_fwd, void* img_bwd,
                           void* stream) {
    if (!w || (!img_fwd && !img_bwd) || tile != 6 || Co < 16 || Ci < 16 || (Co & 15) || (Ci & 15) || (row0 & 15) || (Ct & 15) || row0 < 0 ||
        row0 + Co > Ct || ((uintptr_t)img_fwd & 15) || ((uintptr_t)img_bwd & 15)) return LGD_EINVAL;
    lgd::FilterArgs a{};
    a.w = w; a.scale = scale; a.Co = Co; a.Ci = Ci; a.img_fwd = (char*)img_fwd; a.img_bwd = (char*)img_bwd; a.row0 = row0; a.Ct = Ct;
    lgd::wino6_launch_filter_img(a, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_wino_filter_bwd(const float* dU, long long du_plane, const float* scale, int Co, int Ci, int tile, float* dw, void* stream) {
    if (!dU || !dw || Co < 1 || Ci < 1 || (tile != 4 && tile != 6) || du_plane < (long long)Co * Ci) return LGD_EINVAL;
    lgd::FilterArgs a{};
    a.dU = dU; a.scale = scale; a.dw = dw; a.u_plane = du_plane; a.Co = Co; a.Ci = Ci;
    const long long n = (long long)Co * Ci;
    if (tile == 6) lgd::wino6_launch_filter_bwd(a, (hipStream_t)stream);
    else { LGD_LAUNCH("wino_filter_bwd_kernel", lgd::wino4_filter_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a); }
    return lgd::check_launch();
}

// lgd_wino_filter_bwd (tile 6) over S split-K partials of dU (partial s at dU + s * part_stride), added in fixed order while they are read
int lgd_wino_filter_bwd_parts(const float* dU, long long du_plane, long long part_stride, int S, const float* scale, int Co, int Ci, float* dw,
                              void* stream) {
    if (!dU || !dw || Co < 1 || Ci < 1 || S < 1 || du_plane < (long long)Co * Ci) return LGD_EINVAL;
    lgd::FilterArgs a{};
    a.dU = dU; a.scale = scale; a.dw = dw; a.u_plane = du_plane; a.Co = Co; a.Ci = Ci; a.S = S; a.part_stride = part_stride;
    lgd::wino6_launch_filter_bwd(a, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_wino_in_t_out_t(const float* dV, const void* relu_bits, const int32_t* level_hw_host, int L, int N, int C, int tile, float* dM,
                        void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!dV || !dM || lgd::wino_fill(a, level_hw_host, L, N, C, tile, &blocks) != LGD_OK) return LGD_EINVAL;
    a.buf_in = dV; a.buf_out = dM; a.bits_in = relu_bits;
    if (tile == 6) lgd::wino6_launch_in_t(a, blocks, true, (hipStream_t)stream);
    else { LGD_LAUNCH("wino_in_t_out_t_kernel", lgd::wino4_in_t_kernel<true>, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a); }
    return lgd::check_launch();
}

// ---- F(6x6,3x3) transforms around the f16x2 products of csrc/h2.hip: the frequency buffers they WRITE are split rows (winograd.h)
int lgd_wino_in_h2(const float* const* x_host, const int32_t* level_hw_host, int L, int N, int C, void* V, const float* pre_bias,
                   const float* pre_affine, void* pre_bits, const uint32_t* amax_in, float* inv_out, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!x_host || !V || !amax_in || !inv_out || lgd::wino_fill(a, level_hw_host, L, N, C, 6, &blocks) != LGD_OK) return LGD_EINVAL;
    if ((pre_bits && !pre_bias && !pre_affine) || (pre_bias && pre_affine) || ((uintptr_t)V & 15)) return LGD_EINVAL;
    a.bias = pre_bias; a.pre_affine = pre_affine; a.bits_out = pre_bits;
    for (int l = 0; l < L; ++l) {
        if (!x_host[l]) return LGD_EINVAL;
        a.maps_in[l] = x_host[l];
    }
    a.buf_out = (float*)V; a.h2 = 1; a.amax_in = amax_in; a.scale_out = inv_out;
    lgd::wino6_launch_in(a, blocks, pre_bias != nullptr || pre_affine != nullptr, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_wino_out_t_h2(const float* const* dy_host, const void* relu_bits, const int32_t* level_hw_host, int L, int N, int C, void* dM,
                      const uint32_t* amax_in, float* inv_out64, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!dy_host || !dM || !amax_in || !inv_out64 || ((uintptr_t)dM & 15) || lgd::wino_fill(a, level_hw_host, L, N, C, 6, &blocks) != LGD_OK) return LGD_EINVAL;
    a.bits_in = relu_bits;
    for (int l = 0; l < L; ++l) {
        if (!dy_host[l]) return LGD_EINVAL;
        a.maps_in[l] = dy_host[l];
    }
    a.buf_out = (float*)dM; a.h2 = 1; a.amax_in = amax_in; a.scale_out = inv_out64;
    lgd::wino6_launch_out_t(a, blocks, (hipStream_t)stream);
    return lgd::check_launch();
}

// lgd_wino_out_t_gn writing dM as split rows; *amax_in bounds |ca g - cm - (y - mean) cb| (lgd_h2_gn_bound)
int lgd_wino_out_t_gn_h2(const float* const* g_host, const float* const* y_host, const float* coef, const int32_t* level_hw_host, int L, int N, int C,
                         void* dM, const uint32_t* amax_in, float* inv_out64, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!g_host || !y_host || !coef || !dM || !amax_in || !inv_out64 || ((uintptr_t)dM & 15) || lgd::wino_fill(a, level_hw_host, L, N, C, 6, &blocks) != LGD_OK)
        return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!g_host[l] || !y_host[l]) return LGD_EINVAL;
        a.maps_in[l] = g_host[l];
        a.maps_in2[l] = y_host[l];
    }
    a.gn_coef = coef;
    a.buf_out = (float*)dM; a.h2 = 1; a.amax_in = amax_in; a.scale_out = inv_out64;
    lgd::wino6_launch_out_t(a, blocks, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_wino_in_t_out_t_h2(const float* dV, const void* relu_bits, const int32_t* level_hw_host, int L, int N, int C, void* dM,
                           const uint32_t* bound_in, float* inv_out64, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!dV || !dM || !bound_in || !inv_out64 || ((uintptr_t)dM & 15) || lgd::wino_fill(a, level_hw_host, L, N, C, 6, &blocks) != LGD_OK) return LGD_EINVAL;
    a.buf_in = dV; a.buf_out = (float*)dM; a.bits_in = relu_bits; a.h2 = 1; a.amax_in = bound_in; a.scale_out = inv_out64;
    lgd::wino6_launch_in_t(a, blocks, true, (hipStream_t)stream);
    return lgd::check_launch();
}

// lgd_wino_out / lgd_wino_in_t (tile 6) that also leave max |output| (float bits) in *amax_out: the bound the consumer's f16 scale needs
int lgd_wino_out_amax(const float* M, const float* bias, const int32_t* level_hw_host, int L, int N, int C, int relu, float* const* y_host,
                      void* relu_bits, uint32_t* amax_out, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!M || !y_host || !amax_out || lgd::wino_fill(a, level_hw_host, L, N, C, 6, &blocks) != LGD_OK) return LGD_EINVAL;
    a.bits_out = relu_bits;
    for (int l = 0; l < L; ++l) {
        if (!y_host[l]) return LGD_EINVAL;
        a.maps_out[l] = y_host[l];
    }
    a.buf_in = M; a.bias = bias; a.relu = relu ? 1 : 0; a.amax_out = amax_out;
    lgd::wino6_launch_out(a, blocks, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_wino_in_t_amax(const float* dV, const int32_t* level_hw_host, int L, int N, int C, float* const* dx_host, const void* pre_bits,
                       uint32_t* amax_out, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!dV || !dx_host || !amax_out || lgd::wino_fill(a, level_hw_host, L, N, C, 6, &blocks) != LGD_OK) return LGD_EINVAL;
    a.bits_in = pre_bits;
    for (int l = 0; l < L; ++l) {
        if (!dx_host[l]) return LGD_EINVAL;
        a.maps_out[l] = dx_host[l];
    }
    a.buf_in = dV; a.amax_out = amax_out;
    lgd::wino6_launch_in_t(a, blocks, false, (hipStream_t)stream);
    return lgd::check_launch();
}

int lgd_wino_filter_images_h2(const float* w, const float* scale, int Co, int Ci, int row0, int Ct, void* img_fwd, void* img_bwd,
                              const uint32_t* amax_in, float* inv_out64, void* stream) {
    if (!w || (!img_fwd && !img_bwd) || !amax_in || Co < 16 || Ci < 16 || (Co & 15) || (Ci & 15) || (row0 & 15) || (Ct & 15) || row0 < 0 ||
        row0 + Co > Ct || ((uintptr_t)img_fwd & 15) || ((uintptr_t)img_bwd & 15)) return LGD_EINVAL;
    lgd::FilterArgs a{};
    a.w = w; a.scale = scale; a.Co = Co; a.Ci = Ci; a.img_fwd = (char*)img_fwd; a.img_bwd = (char*)img_bwd; a.row0 = row0; a.Ct = Ct;
    a.amax_in = amax_in; a.inv_out = inv_out64;
    lgd::wino6_launch_filter_img(a, (hipStream_t)stream);
    return lgd::check_launch();
}

}  // extern "C"
