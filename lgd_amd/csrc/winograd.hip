// Winograd F(2x2, 3x3) data transforms for the 3x3 / stride 1 / pad 1 convolutions of the LGD path
//   [ref: dynamic_teacher.py:57,61,67-73 (student_proj_2D, local_inst_proj_2D, refinement_module),
//    models/adapters/sequential_convs.py:10-12, and the student head re-run on the teacher features,
//    distillator.py:107-109 -> retinanet.py:36-43].
// 87 % of the training step was fp32 3x3 convolutions on the library's kernels (100-116 TFLOP/s effective).  The
// minimal-filtering form needs 2.25x fewer multiplies, and its 16 independent (C_out x C_in) x (C_in x tiles) products
// are plain library GEMMs (hipBLASLt via torch.bmm, 100-140 TFLOP/s fp32 MFMA).  What is left is pure HBM streaming,
// which is what these kernels do:
//   wino_in   : x_l (N,C,H_l,W_l)        -> V [C][16][T]       V = B^T d B   per 4x4 input window (stride 2, halo 1)
//   wino_out  : M [C][16][T], bias       -> y_l (N,C,H_l,W_l)  Y = A^T m A   per tile (2x2 outputs) [+bias] [ReLU]
//   wino_out_t: dy_l                     -> dM [C][16][T]      dM = A dy A^T (adjoint of wino_out, weight gradient)
//   wino_in_dual: dy_l -> V(flip) and dM in one pass over dy (the two operands of the backward pass)
// Every module on the path applies ONE filter to all pyramid levels, and a Winograd tile does not care which level it
// came from: the tiles of all L levels are concatenated along T (level l starts at an even offset), so one conv over
// the pyramid is one transform launch + one batched GEMM + one transform launch, and p6/p7 (1.6 % of the pixels, but a
// third of the launches on the per-level library path) ride along for free.  Tile index is the fastest axis everywhere:
// all global accesses are coalesced along x and nothing is transposed.  The frequency planes of one channel are adjacent
// ([C][nf][T]: the GEMM of frequency f sees a (C x T) matrix with row stride nf*T) -- a workgroup's nf output chunks then
// lie within one 45 KB..1.6 MB neighbourhood instead of nf planes 88 MB apart (measured 4.6 -> 5.2 TB/s on the p3 input
// transform, tools/lab/wino4_lab.hip).
// The input gradient is the same pipeline on dy with the 180-degree rotated, (Co,Ci)-transposed filter; the rotation
// is a permutation of the 16 frequencies (flip = 1), so the host reuses U.
#include "common.h"

namespace lgd {

struct WinoArgs {
    const float* maps_in[LGD_MAX_LEVELS];   // per-level NCHW inputs (wino_in / wino_out_t)
    float* maps_out[LGD_MAX_LEVELS];        // per-level NCHW outputs (wino_out)
    const float* mask_ref[LGD_MAX_LEVELS];  // optional: forward outputs y_l; the incoming gradient is zeroed where y <= 0
    const float* buf_in;                    // [16][C][T]
    float* buf_out;                         // [16][C][T]
    float* buf_out2;                        // [16][C][T] (dual)
    const float* bias;
    unsigned short* bits_out;               // optional (tile 4, relu): [C][T] 16-bit ReLU masks of the tiles' 4x4 outputs, bit 4*i+j = y[i][j] > 0
    const unsigned short* bits_in;          // optional (tile 4): the same table as the gradient mask of wino_in / wino_out_t
    long long tile_off[LGD_MAX_LEVELS];     // first tile of the level (even)
    long long T;                            // total tiles incl. per-level padding (even count for tile 2, multiple of 4 for tile 4)
    long long cs;                           // channel stride of the frequency buffers = nf * T  (layout [C][nf][T])
    unsigned blk_off[LGD_MAX_LEVELS + 1];   // first workgroup of the level
    int H[LGD_MAX_LEVELS], W[LGD_MAX_LEVELS], TH[LGD_MAX_LEVELS], TW[LGD_MAX_LEVELS], pair[LGD_MAX_LEVELS];
    int L, N, C, flip, relu;
};

__device__ __forceinline__ int wino_level(const WinoArgs& a) {
    int l = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i)
        if (i < a.L && blockIdx.x >= a.blk_off[i]) l = i;
    return __builtin_amdgcn_readfirstlane(l);
}

// (tx, ty, n) of tile u of a level in 32-bit arithmetic (a level has < 2^31 tiles: wino_fill checks).  The long long form
// `u % TW, (u / TW) % TH, u / (TW * TH)` compiles to four software 64-bit divisions, ~600 of the ~2000 instructions of a transform
// kernel and all of them in front of its first load; the transforms turned out to be as much VALU-issue- as HBM-bound
// (1160 VALU instructions per tile = 87 us of pure issue for the 110 us pyramid launch).
__device__ __forceinline__ void tile_coords(long long u, int TW, int TH, int& tx, int& ty, int& n) {
    const unsigned v = (unsigned)u, r = v / (unsigned)TW, q = r / (unsigned)TH;
    tx = (int)(v - r * (unsigned)TW); ty = (int)(r - q * (unsigned)TH); n = (int)q;
}

// G g G^T of the rotated filter is the frequency permutation 0<->3 (rows 1,2 of G are symmetric under the flip)
__device__ __forceinline__ int freq(int i, int j, int flip) {
    const int pi = (i == 0 || i == 3) ? 3 - i : i, pj = (j == 0 || j == 3) ? 3 - j : j;
    return flip ? 4 * pi + pj : 4 * i + j;
}

// B^T d B with B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
__device__ __forceinline__ void bt4(const float* d, float* o) {
    o[0] = d[0] - d[2]; o[1] = d[1] + d[2]; o[2] = d[2] - d[1]; o[3] = d[1] - d[3];
}

typedef float wino_vf2 __attribute__((ext_vector_type(2)));
typedef float wino_vf4 __attribute__((ext_vector_type(4)));

// V / dM / M are written once and read once by a GEMM that streams 0.7 GB: non-temporal on both sides
// (measured on the p3 input transform: 171 -> 153 us from the store hint alone)
template <int PAIR>
__device__ __forceinline__ void store_freq(float* q, const float (&v)[PAIR]) {
    if constexpr (PAIR == 2) {
        wino_vf2 t; t.x = v[0]; t.y = v[1];
        __builtin_nontemporal_store(t, reinterpret_cast<wino_vf2*>(q));
    } else {
        __builtin_nontemporal_store(v[0], q);
    }
}

// One thread transforms PAIR horizontally adjacent tiles.  PAIR = 2 needs W % 4 == 0: the 6 input columns 4p-1 .. 4p+4
// of a row are one aligned float4 + 2 scalars, and the two tile indices are even/odd neighbours -> float2 stores.
// DUAL also emits dM = A g A^T of the window's 2x2 centre block (rows/cols 1,2 of the window are exactly the tile's
// outputs), so the backward pass reads dy once for both the input-gradient and the weight-gradient operand.
// MASK: the gradient is multiplied by (y > 0) of the forward output (ReLU fused into the producing conv).
template <int PAIR, bool DUAL, bool MASK>
__device__ __forceinline__ void wino_in_body(const WinoArgs& a, int l) {
    const int H = a.H[l], W = a.W[l], TH = a.TH[l], TW = a.TW[l];
    const int TWP = (TW + PAIR - 1) / PAIR;
    const long long units = (long long)a.N * TH * TWP;
    const long long u = (long long)(blockIdx.x - a.blk_off[l]) * 256 + threadIdx.x;
    const int c = blockIdx.y;
    const size_t plane = (size_t)a.T;
    if (u >= units) {
        // an odd tile count is padded by one all-zero tile so that every level starts on an even tile
        if (PAIR == 1 && u == units && (units & 1)) {
            const size_t t = (size_t)a.tile_off[l] + units;
            for (int f = 0; f < 16; ++f) {
                a.buf_out[(size_t)f * plane + (size_t)c * a.cs + t] = 0.f;
                if (DUAL) a.buf_out2[(size_t)f * plane + (size_t)c * a.cs + t] = 0.f;
            }
        }
        return;
    }
    const int txp = (int)(u % TWP), ty = (int)((u / TWP) % TH), n = (int)(u / ((long long)TWP * TH));
    const int tx = txp * PAIR;
    const size_t img = ((size_t)n * a.C + c) * H * W;
    const float* p = a.maps_in[l] + img;
    const float* pm = MASK ? a.mask_ref[l] + img : nullptr;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    constexpr int NC = 2 * PAIR + 2;
    float d[4][NC];
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = y0 + i;
        const bool yok = y >= 0 && y < H;
        const size_t ro = (size_t)(yok ? y : 0) * W;
        const float* row = p + ro;
        if constexpr (PAIR == 2) {
            // the two halo columns are the neighbour lanes' edge values (same image row unless this is the first /
            // last unit of the row, where the halo is the zero padding); only the wave's end lanes load them
            float4 m = yok ? *reinterpret_cast<const float4*>(row + x0 + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (MASK) {
                const float4 k = yok ? *reinterpret_cast<const float4*>(pm + ro + x0 + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
                m.x = k.x > 0.f ? m.x : 0.f; m.y = k.y > 0.f ? m.y : 0.f; m.z = k.z > 0.f ? m.z : 0.f; m.w = k.w > 0.f ? m.w : 0.f;
            }
            const int lane = threadIdx.x & 63;
            float e0 = __shfl_up(m.w, 1), e5 = __shfl_down(m.x, 1);
            if (lane == 0 && txp != 0) {
                e0 = yok ? row[x0] : 0.f;
                if constexpr (MASK) { if (yok) e0 = pm[ro + x0] > 0.f ? e0 : 0.f; }
            }
            if (lane == 63 && txp != TWP - 1) {
                e5 = yok ? row[x0 + 5] : 0.f;
                if constexpr (MASK) { if (yok) e5 = pm[ro + x0 + 5] > 0.f ? e5 : 0.f; }
            }
            if (txp == 0) e0 = 0.f;
            if (txp == TWP - 1) e5 = 0.f;
            d[i][0] = e0; d[i][1] = m.x; d[i][2] = m.y; d[i][3] = m.z; d[i][4] = m.w; d[i][5] = e5;
        } else {
            #pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int x = x0 + j;
                const bool ok = yok && x >= 0 && x < W;
                float e = ok ? row[x] : 0.f;
                if constexpr (MASK) { if (ok) e = pm[ro + x] > 0.f ? e : 0.f; }
                d[i][j] = e;
            }
        }
    }
    const size_t t = (size_t)a.tile_off[l] + ((size_t)n * TH + ty) * TW + tx;
    float* o = a.buf_out + (size_t)c * a.cs + t;
    float v[4][4][PAIR];
    #pragma unroll
    for (int q = 0; q < PAIR; ++q) {
        float r[4][4];
        #pragma unroll
        for (int j = 0; j < 4; ++j) {  // columns: B^T d
            const float col[4] = {d[0][2 * q + j], d[1][2 * q + j], d[2][2 * q + j], d[3][2 * q + j]};
            float w[4];
            bt4(col, w);
            r[0][j] = w[0]; r[1][j] = w[1]; r[2][j] = w[2]; r[3][j] = w[3];
        }
        #pragma unroll
        for (int i = 0; i < 4; ++i) {  // rows: (B^T d) B
            float w[4];
            bt4(r[i], w);
            v[i][0][q] = w[0]; v[i][1][q] = w[1]; v[i][2][q] = w[2]; v[i][3][q] = w[3];
        }
    }
    #pragma unroll
    for (int i = 0; i < 4; ++i)
        #pragma unroll
        for (int j = 0; j < 4; ++j) store_freq<PAIR>(o + (size_t)freq(i, j, a.flip) * plane, v[i][j]);
    if constexpr (DUAL) {
        float* o2 = a.buf_out2 + (size_t)c * a.cs + t;
        float w[4][4][PAIR];
        #pragma unroll
        for (int q = 0; q < PAIR; ++q) {
            // A = [[1,0],[1,1],[1,-1],[0,-1]] on g = window rows/cols 1,2 (outside the map these are the zero halo)
            const float g00 = d[1][2 * q + 1], g01 = d[1][2 * q + 2], g10 = d[2][2 * q + 1], g11 = d[2][2 * q + 2];
            const float r[4][2] = {{g00, g01}, {g00 + g10, g01 + g11}, {g00 - g10, g01 - g11}, {-g10, -g11}};
            #pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[i][0][q] = r[i][0]; w[i][1][q] = r[i][0] + r[i][1]; w[i][2][q] = r[i][0] - r[i][1]; w[i][3][q] = -r[i][1];
            }
        }
        #pragma unroll
        for (int i = 0; i < 4; ++i)
            #pragma unroll
            for (int j = 0; j < 4; ++j) store_freq<PAIR>(o2 + (size_t)(4 * i + j) * plane, w[i][j]);
    }
}

template <bool DUAL, bool MASK>
__global__ __launch_bounds__(256) void wino_in_kernel(WinoArgs a) {
    const int l = wino_level(a);
    if (a.pair[l]) wino_in_body<2, DUAL, MASK>(a, l);
    else wino_in_body<1, DUAL, MASK>(a, l);
}

// Y = A^T m A with A^T = [[1,1,1,0],[0,1,-1,-1]]; PAIR = 2: float2 loads of two neighbouring tiles, float4 row stores
template <int PAIR>
__device__ __forceinline__ void wino_out_body(const WinoArgs& a, int l) {
    const int H = a.H[l], W = a.W[l], TH = a.TH[l], TW = a.TW[l];
    const int TWP = (TW + PAIR - 1) / PAIR;
    const long long u = (long long)(blockIdx.x - a.blk_off[l]) * 256 + threadIdx.x;
    if (u >= (long long)a.N * TH * TWP) return;
    const int c = blockIdx.y;
    const int txp = (int)(u % TWP), ty = (int)((u / TWP) % TH), n = (int)(u / ((long long)TWP * TH));
    const int tx = txp * PAIR;
    const size_t plane = (size_t)a.T;
    const float* m = a.buf_in + (size_t)c * a.cs + (size_t)a.tile_off[l] + ((size_t)n * TH + ty) * TW + tx;
    float q[4][4][PAIR];
    #pragma unroll
    for (int i = 0; i < 4; ++i)
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* s = m + (size_t)freq(i, j, a.flip) * plane;
            if constexpr (PAIR == 2) {
                const wino_vf2 t2 = __builtin_nontemporal_load(reinterpret_cast<const wino_vf2*>(s));
                q[i][j][0] = t2.x; q[i][j][1] = t2.y;
            } else {
                q[i][j][0] = __builtin_nontemporal_load(s);
            }
        }
    const float b = a.bias ? a.bias[c] : 0.f;
    float y[2][2 * PAIR];
    #pragma unroll
    for (int k = 0; k < PAIR; ++k) {
        float r[2][4];
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[0][j] = q[0][j][k] + q[1][j][k] + q[2][j][k];
            r[1][j] = q[1][j][k] - q[2][j][k] - q[3][j][k];
        }
        #pragma unroll
        for (int i = 0; i < 2; ++i) {
            y[i][2 * k] = r[i][0] + r[i][1] + r[i][2] + b;
            y[i][2 * k + 1] = r[i][1] - r[i][2] - r[i][3] + b;
        }
    }
    if (a.relu) {
        #pragma unroll
        for (int i = 0; i < 2; ++i)
            #pragma unroll
            for (int j = 0; j < 2 * PAIR; ++j) y[i][j] = fmaxf(y[i][j], 0.f);
    }
    float* p = a.maps_out[l] + ((size_t)n * a.C + c) * H * W;
    const int oy = 2 * ty, ox = 2 * tx;
    #pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (oy + i >= H) continue;
        float* row = p + (size_t)(oy + i) * W + ox;
        if constexpr (PAIR == 2) {
            *reinterpret_cast<float4*>(row) = make_float4(y[i][0], y[i][1], y[i][2], y[i][3]);
        } else {
            if (ox + 1 < W && ((W & 1) == 0)) *reinterpret_cast<float2*>(row) = make_float2(y[i][0], y[i][1]);
            else { row[0] = y[i][0]; if (ox + 1 < W) row[1] = y[i][1]; }
        }
    }
}

__global__ __launch_bounds__(256) void wino_out_kernel(WinoArgs a) {
    const int l = wino_level(a);
    if (a.pair[l]) wino_out_body<2>(a, l);
    else wino_out_body<1>(a, l);
}

// dM = A dy A^T alone (weight gradient of a conv whose input needs no gradient): the DUAL half of wino_in without V.
__global__ __launch_bounds__(256) void wino_out_t_kernel(WinoArgs a) {
    const int l = wino_level(a);
    const int H = a.H[l], W = a.W[l], TH = a.TH[l], TW = a.TW[l];
    const long long units = (long long)a.N * TH * TW;
    const long long u = (long long)(blockIdx.x - a.blk_off[l]) * 256 + threadIdx.x;
    const int c = blockIdx.y;
    const size_t plane = (size_t)a.T;
    float* o = a.buf_out + (size_t)c * a.cs + (size_t)a.tile_off[l] + u;
    if (u >= units) {
        if (u == units && (units & 1))
            for (int f = 0; f < 16; ++f) o[(size_t)f * plane] = 0.f;
        return;
    }
    const int tx = (int)(u % TW), ty = (int)((u / TW) % TH), n = (int)(u / ((long long)TW * TH));
    const size_t img = ((size_t)n * a.C + c) * H * W;
    const float* p = a.maps_in[l] + img;
    const float* pm = a.mask_ref[l] ? a.mask_ref[l] + img : nullptr;
    const int oy = 2 * ty, ox = 2 * tx;
    float g[2][2];
    #pragma unroll
    for (int i = 0; i < 2; ++i)
        #pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = oy + i < H && ox + j < W;
            const size_t at = (size_t)(oy + i) * W + ox + j;
            float e = ok ? p[at] : 0.f;
            if (ok && pm) e = pm[at] > 0.f ? e : 0.f;
            g[i][j] = e;
        }
    float r[4][2];
    #pragma unroll
    for (int j = 0; j < 2; ++j) { r[0][j] = g[0][j]; r[1][j] = g[0][j] + g[1][j]; r[2][j] = g[0][j] - g[1][j]; r[3][j] = -g[1][j]; }
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v[4] = {r[i][0], r[i][0] + r[i][1], r[i][0] - r[i][1], -r[i][1]};
        #pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v[j], o + (size_t)(4 * i + j) * plane);
    }
}

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

// The nf = 36 values of the workgroup's 256 tiles go through LDS so that every frequency plane is written / read as ONE
// 1 KB run (float4 per lane) instead of 256 B per wave; level tile counts are padded with zero tiles to a multiple of
// kTilePad = 16, so that every level, every frequency plane and every workgroup's runs start on a 64-byte boundary:
// runs that are only 16-byte aligned cost the write-heavy transforms 15 % (tools/lab/wino4_lab.hip, plane stride 8404 vs
// 8400 / 8416 / 8448 floats: 100 vs 85 / 84 / 85 us; 128-byte or 1 KB alignment buys nothing more).  With a pad of 4 the
// 2-image-per-GPU shapes (2,860 tiles) had every plane misaligned.
constexpr int kTilePad = 16;
__device__ __forceinline__ void stage_store36(const float* lds, float* dst, size_t plane, long long t0, long long tend) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    #pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int f = wave + 4 * k;
        const float4 v = *reinterpret_cast<const float4*>(&lds[f * 256 + lane * 4]);
        if (t0 + lane * 4 < tend) {  // tend is a multiple of 4
            wino_vf4 q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
            __builtin_nontemporal_store(q, reinterpret_cast<wino_vf4*>(dst + (size_t)f * plane + lane * 4));
        }
    }
}

// 12 planes (two rows of the 6x6 frequency grid) x 256 tiles: 768 float4, three per thread; f0 = first plane of the slab
#ifndef LGD_WINO_NR
#define LGD_WINO_NR 2
#endif
constexpr int kNR = LGD_WINO_NR;  // frequency rows staged per phase by the input transform

template <int NP = 12>
__device__ __forceinline__ void stage_store12(const float* lds, float* dst, size_t plane, int f0, long long tend) {
    #pragma unroll
    for (int k = 0; k < (NP * 64 + 255) / 256; ++k) {
        const int idx = k * 256 + threadIdx.x, f = idx >> 6, q4 = idx & 63;
        if (idx >= NP * 64) break;
        const float4 v = *reinterpret_cast<const float4*>(&lds[f * 256 + q4 * 4]);
        if (q4 * 4 < tend) {  // tend is a multiple of 4
            wino_vf4 q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
            __builtin_nontemporal_store(q, reinterpret_cast<wino_vf4*>(dst + (size_t)(f0 + f) * plane + q4 * 4));
        }
    }
}

// MASK: 0 = none; 1 = zero the gradient where the forward output y (mask_ref) is <= 0; 2 = the same mask from the 16-bit
// per-tile table the forward output transform wrote (1 bit per pixel instead of re-reading the 4-byte output).
// PRE: the maps are PRE-activations -- the transform reads relu(x + bias[c]) (the bias + ReLU epilogue of the producing 1x1 convolution
// folded into this load: conv1 -> FrozenBN -> ReLU -> conv2 of a bottleneck block) and writes the tile's 16-bit
// activation mask (bit 4*i+j = its own 4x4 block's pixel (i, j) > 0) for the adjoint transform of the backward (wino4_in_t).
template <bool VEC, bool DUAL, int MASK, bool PRE = false>
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
    const size_t img = ((size_t)n * a.C + c) * H * W;
    const float* p = a.maps_in[l] + img;
    const float* pm = MASK == 1 ? a.mask_ref[l] + img : nullptr;
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    const int lane = threadIdx.x & 63;
    const float prb = PRE ? a.bias[c] : 0.f;
    // MASK == 2: this tile's, the upper and the lower tile's masks (rows -1 / 0..3 / 4 of the window), and the same three of
    // the left and right neighbour tiles (columns -1 and 4) from the neighbour lanes; only the wave's end lanes load them
    unsigned mc[3] = {0u, 0u, 0u}, ml[3] = {0u, 0u, 0u}, mr[3] = {0u, 0u, 0u};
    const unsigned short* pb = MASK == 2 ? a.bits_in + (size_t)c * plane + (size_t)a.tile_off[l] : nullptr;
    if constexpr (MASK == 2) {
        const bool up = ty > 0, dn = ty < TH - 1;
        mc[1] = pb[uu];
        mc[0] = up ? pb[uu - TW] : 0u;
        mc[2] = dn ? pb[uu + TW] : 0u;
        #pragma unroll
        for (int k = 0; k < 3; ++k) { ml[k] = wave_shr1(mc[k]); mr[k] = wave_shl1(mc[k]); }
        if (lane == 0 && tx != 0) {
            ml[1] = pb[uu - 1]; ml[0] = up ? pb[uu - TW - 1] : 0u; ml[2] = dn ? pb[uu + TW - 1] : 0u;
        }
        if ((lane == 63 || u + 1 >= units) && tx != TW - 1) {
            mr[1] = pb[uu + 1]; mr[0] = up ? pb[uu - TW + 1] : 0u; mr[2] = dn ? pb[uu + TW + 1] : 0u;
        }
    }
    float d[6][6];
    if constexpr (VEC) {
        // Phase 1: EVERY load of the 6x6 window is issued before anything consumes one -- six aligned float4 rows (+ the ReLU reference
        // rows for MASK == 1) and, on the wave's end lanes, the two halo columns.  Written as one loop (load, mask / activate, DPP
        // halo exchange per row) the compiler put a wait behind each row's load: 6 (plain) to 18 (PRE) exposed HBM latencies per
        // workgroup instead of one (the PRE variant measured 13-17 % slower than the plain one for 36 extra VALU operations).
        float4 m[6], km[6];
        float hl[6], hr[6], kl[6], kr[6];
        const bool needL = lane == 0 && tx != 0, needR = (lane == 63 || u + 1 >= units) && tx != TW - 1;
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int y = y0 + i;
            const bool yok = y >= 0 && y < H;
            const size_t ro = (size_t)(yok ? y : 0) * W;
            m[i] = yok ? *reinterpret_cast<const float4*>(p + ro + x0 + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (MASK == 1) km[i] = yok ? *reinterpret_cast<const float4*>(pm + ro + x0 + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (needL) {
            #pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int y = y0 + i;
                const bool yok = y >= 0 && y < H;
                const size_t ro = (size_t)(yok ? y : 0) * W;
                hl[i] = yok ? p[ro + x0] : 0.f;
                if constexpr (MASK == 1) kl[i] = yok ? pm[ro + x0] : 0.f;
            }
        }
        if (needR) {
            #pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int y = y0 + i;
                const bool yok = y >= 0 && y < H;
                const size_t ro = (size_t)(yok ? y : 0) * W;
                hr[i] = yok ? p[ro + x0 + 5] : 0.f;
                if constexpr (MASK == 1) kr[i] = yok ? pm[ro + x0 + 5] : 0.f;
            }
        }
        // Phase 2: masks / folded activation, halo exchange, window assembly
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int y = y0 + i;
            const bool yok = y >= 0 && y < H;
            // window row i lives in tile-row r of family f (0: upper tile, 1: this tile, 2: lower tile)
            const int f = i == 0 ? 0 : (i == 5 ? 2 : 1), r = i == 0 ? 3 : (i == 5 ? 0 : i - 1);
            float4 v = m[i];
            if constexpr (MASK == 1) {
                v.x = km[i].x > 0.f ? v.x : 0.f; v.y = km[i].y > 0.f ? v.y : 0.f; v.z = km[i].z > 0.f ? v.z : 0.f; v.w = km[i].w > 0.f ? v.w : 0.f;
            }
            if constexpr (MASK == 2) {
                const unsigned nib = mc[f] >> (4 * r);
                v.x = (nib & 1u) ? v.x : 0.f; v.y = (nib & 2u) ? v.y : 0.f; v.z = (nib & 4u) ? v.z : 0.f; v.w = (nib & 8u) ? v.w : 0.f;
            }
            // PRE: rows beyond the map stay the zero padding of the ACTIVATION: relu(0 + -inf) = 0, no branch
            const float pbv = PRE ? (yok ? prb : -INFINITY) : 0.f;
            if constexpr (PRE) { v.x = fmaxf(v.x + pbv, 0.f); v.y = fmaxf(v.y + pbv, 0.f); v.z = fmaxf(v.z + pbv, 0.f); v.w = fmaxf(v.w + pbv, 0.f); }
            // halo columns = the neighbour lanes' edge values (same image row unless first / last tile of the row): one DPP move each
            // (measured equal to ds_bpermute shuffles, 136.1 vs 136.4 us)
            float e0 = wave_shr1(v.w), e5 = wave_shl1(v.x);
            if (needL) {
                e0 = hl[i];
                if constexpr (MASK == 1) e0 = kl[i] > 0.f ? e0 : 0.f;
                if constexpr (MASK == 2) e0 = ((ml[f] >> (4 * r + 3)) & 1u) ? e0 : 0.f;
                if constexpr (PRE) e0 = fmaxf(e0 + pbv, 0.f);
            }
            if (needR) {
                e5 = hr[i];
                if constexpr (MASK == 1) e5 = kr[i] > 0.f ? e5 : 0.f;
                if constexpr (MASK == 2) e5 = ((mr[f] >> (4 * r)) & 1u) ? e5 : 0.f;
                if constexpr (PRE) e5 = fmaxf(e5 + pbv, 0.f);
            }
            if (tx == 0) e0 = 0.f;
            if (tx == TW - 1) e5 = 0.f;
            d[i][0] = e0; d[i][1] = v.x; d[i][2] = v.y; d[i][3] = v.z; d[i][4] = v.w; d[i][5] = e5;
        }
    } else {
        // W % 4 != 0 (res5 / p5 / p7 at 800x1344): 36 dword loads, all issued before the first is consumed (same reason as above)
        float kk[6][6];
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
                if constexpr (MASK == 1) kk[i][j] = ok ? pm[ro + x] : 0.f;
            }
        }
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int y = y0 + i;
            const bool yok = y >= 0 && y < H;
            const int f = i == 0 ? 0 : (i == 5 ? 2 : 1), r = i == 0 ? 3 : (i == 5 ? 0 : i - 1);
            #pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int x = x0 + j;
                const bool ok = yok && x >= 0 && x < W;
                float e = d[i][j];
                if constexpr (MASK == 1) e = kk[i][j] > 0.f ? e : 0.f;
                if constexpr (MASK == 2) {
                    const unsigned w16 = j == 0 ? ml[f] : (j == 5 ? mr[f] : mc[f]);
                    const int cc = j == 0 ? 3 : (j == 5 ? 0 : j - 1);
                    e = ((w16 >> (4 * r + cc)) & 1u) ? e : 0.f;
                }
                if constexpr (PRE) e = fmaxf(e + (ok ? prb : -INFINITY), 0.f);
                d[i][j] = e;
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
            a.bits_out[(size_t)c * plane + (size_t)a.tile_off[l] + u] = (unsigned short)bits;
        }
    }
    {
        float r[6][6];
        #pragma unroll
        for (int j = 0; j < 6; ++j) {  // columns: B^T d
            const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
            float w[6];
            bt6(col, w);
            #pragma unroll
            for (int i = 0; i < 6; ++i) r[i][j] = w[i];
        }
        // rows: (B^T d) B, staged two frequency rows (12 planes, 12 KB) at a time: 3x the resident workgroups of a
        // 36-plane slab (4 % faster on the p3 transform, tools/lab/wino4_lab.hip)
        #pragma unroll
        for (int ph = 0; ph < 6 / kNR; ++ph) {
            if (ph) __syncthreads();
            #pragma unroll
            for (int ii = 0; ii < kNR; ++ii) {
                float w[6];
                bt6(r[kNR * ph + ii], w);
                #pragma unroll
                for (int j = 0; j < 6; ++j) lds[(6 * ii + j) * 256 + threadIdx.x] = on ? w[j] : 0.f;
            }
            __syncthreads();
            stage_store12<6 * kNR>(lds, a.buf_out + base, plane, 6 * kNR * ph, tend);
        }
    }
    if constexpr (DUAL) {
        // dM = A g A^T with g = the tile's own 4x4 block = window rows/cols 1..4 (zero beyond the map)
        float r[6][4];
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float col[4] = {d[1][j + 1], d[2][j + 1], d[3][j + 1], d[4][j + 1]};
            float w[6];
            a6(col, w);
            #pragma unroll
            for (int i = 0; i < 6; ++i) r[i][j] = w[i];
        }
        #pragma unroll
        for (int ph = 0; ph < 6 / kNR; ++ph) {
            __syncthreads();
            #pragma unroll
            for (int ii = 0; ii < kNR; ++ii) {
                float w[6];
                a6(r[kNR * ph + ii], w);
                #pragma unroll
                for (int j = 0; j < 6; ++j) lds[(6 * ii + j) * 256 + threadIdx.x] = on ? w[j] : 0.f;
            }
            __syncthreads();
            stage_store12<6 * kNR>(lds, a.buf_out2 + base, plane, 6 * kNR * ph, tend);
        }
    }
}

template <bool DUAL, int MASK, bool PRE = false>
__global__ __launch_bounds__(256) void wino4_in_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[12 * 256];
    const int l = wino_level(a);
    if (a.pair[l]) wino4_in_body<true, DUAL, MASK, PRE>(a, l, lds);
    else wino4_in_body<false, DUAL, MASK, PRE>(a, l, lds);
}

template <bool VEC, bool STAGE>
__device__ __forceinline__ void wino4_out_body(const WinoArgs& a, int l, float* lds) {
    const int H = a.H[l], W = a.W[l], TH = a.TH[l], TW = a.TW[l];
    const long long units = (long long)a.N * TH * TW, padded = (units + kTilePad - 1) & ~(long long)(kTilePad - 1);
    const long long t0 = (long long)(blockIdx.x - a.blk_off[l]) * 256;
    const long long u = t0 + threadIdx.x;
    const int c = blockIdx.y;
    const size_t plane = (size_t)a.T;
    const float* m = a.buf_in + (size_t)c * a.cs + (size_t)a.tile_off[l] + t0;
    float mm[6][6];
    if constexpr (STAGE) {   // 1 KB runs of two frequency rows (12 planes, 12 KB) at a time through LDS
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
    }
    if (u >= units) return;
    if constexpr (!STAGE) {
        #pragma unroll
        for (int i = 0; i < 6; ++i)
            #pragma unroll
            for (int j = 0; j < 6; ++j) mm[i][j] = __builtin_nontemporal_load(m + (size_t)(6 * i + j) * plane + threadIdx.x);
    }
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
        if (i == 3 && a.bits_out) a.bits_out[(size_t)c * plane + (size_t)a.tile_off[l] + u] = (unsigned short)bits;
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

template <bool STAGE>
__global__ __launch_bounds__(256) void wino4_out_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[STAGE ? 12 * 256 : 4];
    const int l = wino_level(a);
    if (a.pair[l]) wino4_out_body<true, STAGE>(a, l, lds);
    else wino4_out_body<false, STAGE>(a, l, lds);
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
    const size_t img = ((size_t)n * a.C + c) * H * W;
    const float* p = a.maps_in[l] + img;
    const float* pm = a.mask_ref[l] ? a.mask_ref[l] + img : nullptr;
    const unsigned mb = a.bits_in ? a.bits_in[(size_t)c * plane + (size_t)a.tile_off[l] + uu] : 0xffffu;
    // all loads of the 4x4 block first, then the masks: with load + (runtime-optional) reference load + mask per row in one loop the
    // compiler waited for each row before issuing the next (four exposed HBM latencies per workgroup instead of one)
    float g[4][4], kref[4][4];
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
    if (pm) {
        #pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = 4 * ty + i;
            const bool yok = y < H;
            const size_t ro = (size_t)(yok ? y : 0) * W + 4 * tx;
            if constexpr (VEC) {
                const float4 k = yok ? *reinterpret_cast<const float4*>(pm + ro) : make_float4(0.f, 0.f, 0.f, 0.f);
                kref[i][0] = k.x; kref[i][1] = k.y; kref[i][2] = k.z; kref[i][3] = k.w;
            } else {
                #pragma unroll
                for (int j = 0; j < 4; ++j) kref[i][j] = (yok && 4 * tx + j < W) ? pm[ro + j] : 0.f;
            }
        }
        #pragma unroll
        for (int i = 0; i < 4; ++i)
            #pragma unroll
            for (int j = 0; j < 4; ++j) g[i][j] = kref[i][j] > 0.f ? g[i][j] : 0.f;
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
        stage_store12<12>(lds, dst, plane, 12 * ph, padded - t0);
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
        const unsigned mb = a.bits_in ? a.bits_in[(size_t)c * plane + (size_t)a.tile_off[l] + u] : 0xffffu;
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
        const unsigned mb = a.bits_in ? a.bits_in[(size_t)c * plane + (size_t)a.tile_off[l] + uu] : 0xffffu;
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
            stage_store12<12>(lds, dst, plane, 12 * ph, padded - t0);
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

struct FilterArgs {
    const float* w; const float* scale; const float* dU;
    float* U; float* Ut; float* dw;
    long long u_plane, ut_plane, ut_ld;
    int Co, Ci;
};

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

static long long level_tiles(int N, int H, int W, int tile) {
    if (tile == 4) return (((long long)N * ((H + 3) / 4) * ((W + 3) / 4)) + kTilePad - 1) & ~(long long)(kTilePad - 1);
    const long long t = (long long)N * ((H + 1) / 2) * ((W + 1) / 2);
    return t + (t & 1);
}

// fills the per-level tables; mode 0: pair units where W % 4 == 0 (wino_in / wino_out), 1: one tile per thread
static int wino_fill(WinoArgs& a, const int32_t* level_hw, int L, int N, int C, int flip, int mode, int tile, unsigned* blocks) {
    if (!level_hw || L < 1 || L > LGD_MAX_LEVELS || N < 1 || C < 1 || C > 65535 || (tile != 2 && tile != 4) || (tile == 4 && flip))
        return LGD_EINVAL;
    a.L = L; a.N = N; a.C = C; a.flip = flip ? 1 : 0; a.relu = 0;
    a.bias = nullptr; a.buf_in = nullptr; a.buf_out = a.buf_out2 = nullptr; a.bits_out = nullptr; a.bits_in = nullptr;
    long long off = 0;
    unsigned blk = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.maps_in[l] = nullptr; a.maps_out[l] = nullptr; a.mask_ref[l] = nullptr;
        a.H[l] = a.W[l] = a.TH[l] = a.TW[l] = a.pair[l] = 0; a.tile_off[l] = 0; a.blk_off[l] = 0;
    }
    for (int l = 0; l < L; ++l) {
        const int H = level_hw[2 * l], W = level_hw[2 * l + 1];
        if (H < 1 || W < 1) return LGD_EINVAL;
        a.H[l] = H; a.W[l] = W; a.TH[l] = (H + tile - 1) / tile; a.TW[l] = (W + tile - 1) / tile;
        a.pair[l] = (mode == 0 && W % 4 == 0) ? 1 : 0;  // tile 2: two tiles per thread; tile 4: aligned float4 rows
        a.tile_off[l] = off;
        a.blk_off[l] = blk;
        const long long units = (long long)N * a.TH[l] * ((tile == 2 && a.pair[l]) ? a.TW[l] / 2 : a.TW[l]);
        if (units >= (1LL << 31) - 256) return LGD_EINVAL;   // the kernels index a level's tiles in 32 bits
        blk += (unsigned)((units + 1 + 255) / 256);  // +1: the thread that writes the zero pad tile (tile 2)
        off += level_tiles(N, H, W, tile);
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
    if (!level_hw_host || L < 1 || N < 1 || (tile != 2 && tile != 4)) return 0;
    long long t = 0;
    for (int l = 0; l < L; ++l) t += lgd::level_tiles(N, level_hw_host[2 * l], level_hw_host[2 * l + 1], tile);
    return (size_t)t;
}

int lgd_wino_in(const float* const* x_host, const float* const* relu_ref_host, const uint16_t* relu_bits, const int32_t* level_hw_host,
                int L, int N, int C, int tile, int flip, float* V, float* dM, const float* pre_bias, uint16_t* pre_bits, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!x_host || !V || lgd::wino_fill(a, level_hw_host, L, N, C, flip, 0, tile, &blocks) != LGD_OK) return LGD_EINVAL;
    if (relu_bits && (relu_ref_host || tile != 4)) return LGD_EINVAL;
    if ((pre_bias || pre_bits) && (!pre_bias || tile != 4 || dM || relu_bits || relu_ref_host)) return LGD_EINVAL;
    a.bits_in = relu_bits;
    a.bias = pre_bias; a.bits_out = pre_bits;
    for (int l = 0; l < L; ++l) {
        if (!x_host[l] || (relu_ref_host && !relu_ref_host[l])) return LGD_EINVAL;
        a.maps_in[l] = x_host[l];
        if (relu_ref_host) a.mask_ref[l] = relu_ref_host[l];
    }
    a.buf_out = V; a.buf_out2 = dM;
    const dim3 grid(blocks, C), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool mask = relu_ref_host != nullptr;
    if (tile == 4 && pre_bias) {
        LGD_LAUNCH("wino_in_kernel", (lgd::wino4_in_kernel<false, 0, true>), grid, block, 0, st, a);
    } else if (tile == 4) {
        if (dM) {
            if (relu_bits) { LGD_LAUNCH("wino_in_dual_kernel", (lgd::wino4_in_kernel<true, 2>), grid, block, 0, st, a); }
            else if (mask) { LGD_LAUNCH("wino_in_dual_kernel", (lgd::wino4_in_kernel<true, 1>), grid, block, 0, st, a); }
            else { LGD_LAUNCH("wino_in_dual_kernel", (lgd::wino4_in_kernel<true, 0>), grid, block, 0, st, a); }
        } else {
            if (relu_bits) { LGD_LAUNCH("wino_in_kernel", (lgd::wino4_in_kernel<false, 2>), grid, block, 0, st, a); }
            else if (mask) { LGD_LAUNCH("wino_in_kernel", (lgd::wino4_in_kernel<false, 1>), grid, block, 0, st, a); }
            else { LGD_LAUNCH("wino_in_kernel", (lgd::wino4_in_kernel<false, 0>), grid, block, 0, st, a); }
        }
    } else {
        if (dM) {
            if (mask) { LGD_LAUNCH("wino_in_dual_kernel", (lgd::wino_in_kernel<true, true>), grid, block, 0, st, a); }
            else { LGD_LAUNCH("wino_in_dual_kernel", (lgd::wino_in_kernel<true, false>), grid, block, 0, st, a); }
        } else {
            if (mask) { LGD_LAUNCH("wino_in_kernel", (lgd::wino_in_kernel<false, true>), grid, block, 0, st, a); }
            else { LGD_LAUNCH("wino_in_kernel", (lgd::wino_in_kernel<false, false>), grid, block, 0, st, a); }
        }
    }
    return lgd::check_launch();
}

int lgd_wino_out(const float* M, const float* bias, const int32_t* level_hw_host, int L, int N, int C, int tile, int flip,
                 int relu, float* const* y_host, uint16_t* relu_bits, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!M || !y_host || lgd::wino_fill(a, level_hw_host, L, N, C, flip, 0, tile, &blocks) != LGD_OK) return LGD_EINVAL;
    if (relu_bits && tile != 4) return LGD_EINVAL;
    a.bits_out = relu_bits;
    for (int l = 0; l < L; ++l) {
        if (!y_host[l]) return LGD_EINVAL;
        a.maps_out[l] = y_host[l];
    }
    a.buf_in = M; a.bias = bias; a.relu = relu ? 1 : 0;
    // reads of M staged through LDS two frequency rows (12 KB) at a time: 1 KB runs instead of 256 B per wave, measured
    // 83 -> 77 us in the step (HBM-cold 109 -> 100 us = 6.0 TB/s); a 36-plane slab (36 KB, a third of the resident
    // workgroups) was slower than the direct loads (124 us)
    if (tile == 4) { LGD_LAUNCH("wino_out_kernel", lgd::wino4_out_kernel<true>, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a); }
    else { LGD_LAUNCH("wino_out_kernel", lgd::wino_out_kernel, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a); }
    return lgd::check_launch();
}

int lgd_wino_out_t(const float* const* dy_host, const float* const* relu_ref_host, const uint16_t* relu_bits,
                   const int32_t* level_hw_host, int L, int N, int C, int tile, float* dM, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!dy_host || !dM || lgd::wino_fill(a, level_hw_host, L, N, C, 0, tile == 4 ? 0 : 1, tile, &blocks) != LGD_OK) return LGD_EINVAL;
    if (relu_bits && (relu_ref_host || tile != 4)) return LGD_EINVAL;
    a.bits_in = relu_bits;
    for (int l = 0; l < L; ++l) {
        if (!dy_host[l] || (relu_ref_host && !relu_ref_host[l])) return LGD_EINVAL;
        a.maps_in[l] = dy_host[l];
        if (relu_ref_host) a.mask_ref[l] = relu_ref_host[l];
    }
    a.buf_out = dM;
    if (tile == 4) { LGD_LAUNCH("wino_out_t_kernel", lgd::wino4_out_t_kernel, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a); }
    else { LGD_LAUNCH("wino_out_t_kernel", lgd::wino_out_t_kernel, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a); }
    return lgd::check_launch();
}

int lgd_wino_in_t(const float* dV, const int32_t* level_hw_host, int L, int N, int C, int tile, float* const* dx_host,
                  const uint16_t* pre_bits, void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!dV || !dx_host || tile != 4 || lgd::wino_fill(a, level_hw_host, L, N, C, 0, 0, tile, &blocks) != LGD_OK) return LGD_EINVAL;
    a.bits_in = pre_bits;
    for (int l = 0; l < L; ++l) {
        if (!dx_host[l]) return LGD_EINVAL;
        a.maps_out[l] = dx_host[l];
    }
    a.buf_in = dV;
    LGD_LAUNCH("wino_in_t_kernel", lgd::wino4_in_t_kernel<false>, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

int lgd_wino_filter_fwd(const float* w, const float* scale, int Co, int Ci, float* U, long long u_plane, float* Ut, long long ut_ld,
                        long long ut_plane, void* stream) {
    if (!w || !U || !Ut || Co < 1 || Ci < 1 || u_plane < (long long)Co * Ci || ut_ld < Co || ut_plane < (long long)Ci * ut_ld) return LGD_EINVAL;
    lgd::FilterArgs a{};
    a.w = w; a.scale = scale; a.U = U; a.Ut = Ut; a.u_plane = u_plane; a.ut_plane = ut_plane; a.ut_ld = ut_ld; a.Co = Co; a.Ci = Ci;
    LGD_LAUNCH("wino_filter_kernel", lgd::wino4_filter_fwd_kernel, dim3((Ci + 15) / 16, (Co + 15) / 16), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

int lgd_wino_filter_bwd(const float* dU, long long du_plane, const float* scale, int Co, int Ci, float* dw, void* stream) {
    if (!dU || !dw || Co < 1 || Ci < 1 || du_plane < (long long)Co * Ci) return LGD_EINVAL;
    lgd::FilterArgs a{};
    a.dU = dU; a.scale = scale; a.dw = dw; a.u_plane = du_plane; a.Co = Co; a.Ci = Ci;
    const long long n = (long long)Co * Ci;
    LGD_LAUNCH("wino_filter_bwd_kernel", lgd::wino4_filter_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

int lgd_wino_in_t_out_t(const float* dV, const uint16_t* relu_bits, const int32_t* level_hw_host, int L, int N, int C, int tile, float* dM,
                        void* stream) {
    lgd::WinoArgs a;
    unsigned blocks;
    if (!dV || !dM || tile != 4 || lgd::wino_fill(a, level_hw_host, L, N, C, 0, 0, tile, &blocks) != LGD_OK) return LGD_EINVAL;
    a.buf_in = dV; a.buf_out = dM; a.bits_in = relu_bits;
    LGD_LAUNCH("wino_in_t_out_t_kernel", lgd::wino4_in_t_kernel<true>, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
