// Shared pieces of the Winograd data-transform kernels (winograd.hip: F(4x4,3x3); winograd6.hip: F(6x6,3x3)).
#pragma once
#include <type_traits>

#include "common.h"

namespace lgd {

struct WinoArgs {
    const float* maps_in[LGD_MAX_LEVELS];   // per-level NCHW inputs (wino_in / wino_out_t)
    float* maps_out[LGD_MAX_LEVELS];        // per-level NCHW outputs (wino_out)
    const float* maps_in2[LGD_MAX_LEVELS];  // optional (wino_out_t, with gn_coef): the convolution's own forward outputs y
    const float* gn_coef;                   // optional (wino_out_t): [L][N][C][4] (ca, cm, mu, cb) of the GroupNorm that follows the convolution --
                                            //   maps_in are then gradients w.r.t. the GroupNorm OUTPUT and dy = ca * g - cm - (y - mu) * cb
    const float* buf_in;                    // [C][nf][T]
    float* buf_out;                         // [C][nf][T]
    const float* bias;
    const float* pre_affine;                // optional (input transform, PRE): [L][N][C][2] scale, shift -- the maps are the inputs of a per-(map, sample,
                                            // channel) affine + ReLU (GroupNorm + ReLU with its statistics folded): relu(x * scale + shift)
    void* bits_out;                         // optional: [C][T] per-tile activation masks, bit tile*i+j = pixel (i, j) of the tile's block > 0
    const void* bits_in;                    //   (uint16 per 4x4 tile, uint64 per 6x6 tile); the same table read as the gradient mask
    long long tile_off[LGD_MAX_LEVELS];     // first tile of the level (multiple of kTilePad)
    long long T;                            // total tiles incl. per-level padding
    long long cs;                           // channel stride of the frequency buffers = nf * T  (layout [C][nf][T])
    unsigned blk_off[LGD_MAX_LEVELS + 1];   // first workgroup of the level
    int H[LGD_MAX_LEVELS], W[LGD_MAX_LEVELS], TH[LGD_MAX_LEVELS], TW[LGD_MAX_LEVELS], pair[LGD_MAX_LEVELS];
    int L, N, C, relu;
    // f16x2 split frequency buffers (csrc/h2.hip; F(6x6,3x3) only).  h2 != 0: the transform that WRITES a frequency buffer (wino6_in, wino6_out_t,
    // the fused backward link) writes it as split rows -- (h, m) f16 pairs of x * 2^e, layout h2_piece_off() -- with e derived from *amax_in, an
    // upper bound (float bits) of the magnitude of the transform's input AFTER its pre-activation / mask; thread 0 of workgroup (0, 0) records
    // the inverse scale(s) 2^-e in scale_out: 1 float (wino6_in) or 64 floats (wino6_out_t / link: per frequency).
    const unsigned* amax_in;
    float* scale_out;
    unsigned* amax_out;                     // optional (wino6_out, wino6_in_t): atomicMax of the float bits of |output| (pre-zeroed by the caller)
    int h2;
};

// ---- f16x2 split rows: element t of a row (a frequency plane of one channel: T tiles, 4 T bytes) -- blocks of kH2Block tiles, the
// block's h values (2 kH2Block bytes) followed by its m values
constexpr int kH2Block = 32;
__host__ __device__ inline long long h2_piece_off(long long t, int piece) {   // bytes from the row start
    return (t / kH2Block) * (4 * kH2Block) + piece * (2 * kH2Block) + (t % kH2Block) * 2;
}
// power-of-two multiplier 2^e with bound * 2^(lg_gain) * 2^e < 2^15 (f16: max 65504), bound given by its float bits; and its exponent
__device__ __forceinline__ int h2_exponent(unsigned bound_bits, int lg_gain) {
    const int ex = (int)((bound_bits >> 23) & 0xffu) - 127;   // bound < 2^(ex + 1)
    const int e = 14 - ex - lg_gain;
    return e < -126 ? -126 : (e > 126 ? 126 : e);
}
__device__ __forceinline__ float h2_pow2(int e) {
    e = e < -126 ? -126 : (e > 127 ? 127 : e);
    return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}
// x (already scaled) -> h | m << 16
__device__ __forceinline__ uint32_t h2_pack(float v) {
    typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));
    typedef float h2_f32x2 __attribute__((ext_vector_type(2)));
    // (v must exist as ONE fp32 value: fused into its producer -- v_fma_mixlo_f16 rounds a * b + c to f16 in a single step -- the h that r is taken
    //  against could differ by an f16 ulp from the f16(v) the packed conversion below stores)
    asm volatile("" : "+v"(v));
    const _Float16 h = (_Float16)v;
    const float r = v - (float)h;
    const h2_f16x2 pk = __builtin_convertvector((h2_f32x2){v, r}, h2_f16x2);   // v_cvt_pk_f16_f32
    return __builtin_bit_cast(uint32_t, pk);
}
// upper bounds (as exponents) of the abs row sums of the transform matrices: B^T (input transform; uniform 4 per dimension: 15 < 2^4) and A (6 -> 8,
// adjoint output transform: row sums 1, 6, 6, 63, 63, 1.97, 1.97, 1)
constexpr int kH2LgBt = 4;
__host__ __device__ constexpr int h2_lg_a(int i) { return i == 0 || i == 7 ? 0 : (i == 1 || i == 2 ? 3 : (i == 3 || i == 4 ? 6 : 1)); }

// NP planes x 256 tiles of packed (h | m << 16) words staged in LDS ([plane][tile]) -> split rows.  Thread <-> (plane, octet of 8 tiles): two 16-byte
// LDS reads, the halves sorted by v_perm, one 16-byte store per piece.  dst: the word of the workgroup's first tile in plane 0 of the channel (tile
// index = word index: a row holds 4 bytes per tile in either format); t0: that tile's index in the row (a multiple of 16).
typedef uint32_t wino_vu4 __attribute__((ext_vector_type(4)));
template <int NP>
__device__ __forceinline__ void stage_store_h2(const uint32_t* lds, uint32_t* dst, size_t plane, int f0, long long tend, long long t0) {
    #pragma unroll
    for (int k = 0; k < NP * 32 / 256; ++k) {
        const int idx = k * 256 + threadIdx.x, f = idx >> 5, o = idx & 31;
        const wino_vu4 a = *reinterpret_cast<const wino_vu4*>(&lds[f * 256 + o * 8]);
        const wino_vu4 b = *reinterpret_cast<const wino_vu4*>(&lds[f * 256 + o * 8 + 4]);
        wino_vu4 h, m;
        h.x = __builtin_amdgcn_perm(a.y, a.x, 0x05040100u); m.x = __builtin_amdgcn_perm(a.y, a.x, 0x07060302u);
        h.y = __builtin_amdgcn_perm(a.w, a.z, 0x05040100u); m.y = __builtin_amdgcn_perm(a.w, a.z, 0x07060302u);
        h.z = __builtin_amdgcn_perm(b.y, b.x, 0x05040100u); m.z = __builtin_amdgcn_perm(b.y, b.x, 0x07060302u);
        h.w = __builtin_amdgcn_perm(b.w, b.z, 0x05040100u); m.w = __builtin_amdgcn_perm(b.w, b.z, 0x07060302u);
        if (o * 8 < tend) {  // tend is a multiple of 32
            char* row = reinterpret_cast<char*>(dst + (size_t)(f0 + f) * plane) - 4 * t0;   // the row's first byte
            const long long t = t0 + o * 8;
            __builtin_nontemporal_store(h, reinterpret_cast<wino_vu4*>(row + h2_piece_off(t, 0)));
            __builtin_nontemporal_store(m, reinterpret_cast<wino_vu4*>(row + h2_piece_off(t, 1)));
        }
    }
}

__device__ __forceinline__ int wino_level(const WinoArgs& a) {
    int l = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i)
        if (i < a.L && blockIdx.x >= a.blk_off[i]) l = i;
    return __builtin_amdgcn_readfirstlane(l);
}

// (tx, ty, n) of tile u of a level in 32-bit arithmetic (a level has < 2^31 tiles: wino_fill checks).  The long long form
// `u % TW, (u / TW) % TH, u / (TW * TH)` compiles to four software 64-bit divisions, ~600 of the ~2000 instructions of a transform
// kernel and all of them in front of its first load.
__device__ __forceinline__ void tile_coords(long long u, int TW, int TH, int& tx, int& ty, int& n) {
    const unsigned v = (unsigned)u, r = v / (unsigned)TW, q = r / (unsigned)TH;
    tx = (int)(v - r * (unsigned)TW); ty = (int)(r - q * (unsigned)TH); n = (int)q;
}

typedef float wino_vf2 __attribute__((ext_vector_type(2)));
typedef float wino_vf4 __attribute__((ext_vector_type(4)));

// Level tile counts are padded with zero tiles to a multiple of kTilePad = 32 (16 until round 4; 32 = kH2Block: a split row is whole blocks), so that every level, every frequency plane and every
// workgroup's runs start on a 64-byte boundary: runs that are only 16-byte aligned cost the write-heavy transforms 15 %
// (tools/lab/wino4_lab.hip, plane stride 8404 vs 8400 / 8416 / 8448 floats: 100 vs 85 / 84 / 85 us).
constexpr int kTilePad = 32;

// NP planes x 256 tiles staged in LDS ([plane][tile]) -> NP runs of 1 KB (float4 per lane), non-temporal: V / dM / M are written
// once and read once by a GEMM that streams 0.7 GB.  f0 = first plane of the slab, tend = tiles of this workgroup that exist.
template <int NP>
__device__ __forceinline__ void stage_store(const float* lds, float* dst, size_t plane, int f0, long long tend) {
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

// host side (winograd.hip)
long long wino_level_tiles(int N, int H, int W, int tile);
int wino_fill(WinoArgs& a, const int32_t* level_hw, int L, int N, int C, int tile, unsigned* blocks);

// F(6x6,3x3) launches (winograd6.hip); `a` filled by wino_fill(tile = 6)
void wino6_launch_in(const WinoArgs& a, unsigned blocks, bool pre, hipStream_t st);
void wino6_launch_out(const WinoArgs& a, unsigned blocks, hipStream_t st);
void wino6_launch_out_t(const WinoArgs& a, unsigned blocks, hipStream_t st);   // a.gn_coef: the GroupNorm-backward form
void wino6_launch_in_t(const WinoArgs& a, unsigned blocks, bool fuse, hipStream_t st);
struct FilterArgs {
    const float* w; const float* scale; const float* dU;
    float* U; float* Ut; float* dw;
    long long u_plane, ut_plane, ut_ld;
    int Co, Ci;
    char* img_fwd; char* img_bwd; int row0, Ct;   // wino6_filter_img_kernel: gemm3 operand images of the stacked filter (rows row0 .. row0 + Co of Ct)
    int S; long long part_stride;                 // wino6_filter_bwd_kernel: dU as S split-K partials (0 / 1: plain)
    const unsigned* amax_in; float* inv_out;      // f16x2 images (csrc/h2.hip): bound of |w . scale| (float bits) in, the 64 inverse scales out
};
void wino6_launch_filter_fwd(const FilterArgs& a, hipStream_t st);
void wino6_launch_filter_img(const FilterArgs& a, hipStream_t st);
void wino6_launch_filter_bwd(const FilterArgs& a, hipStream_t st);

}  // namespace lgd
