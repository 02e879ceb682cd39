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
// kernel and all of them in front of its first load.
__device__ __forceinline__ void tile_coords(long long u, int TW, int TH, int& tx, int& ty, int& n) {
    const unsigned v = (unsigned)u, r = v / (unsigned)TW, q = r / (unsigned)TH;
    tx = (int)(v - r * (unsigned)TW); ty = (int)(r - q * (unsigned)TH); n = (int)q;
}

typedef float wino_vf2 __attribute__((ext_vector_type(2)));
typedef float wino_vf4 __attribute__((ext_vector_type(4)));

// Level tile counts are padded with zero tiles to a multiple of kTilePad = 16, so that every level, every frequency plane and every
// workgroup's runs start on a 64-byte boundary: runs that are only 16-byte aligned cost the write-heavy transforms 15 %
// (tools/lab/wino4_lab.hip, plane stride 8404 vs 8400 / 8416 / 8448 floats: 100 vs 85 / 84 / 85 us).
constexpr int kTilePad = 16;

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
};
void wino6_launch_filter_fwd(const FilterArgs& a, hipStream_t st);
void wino6_launch_filter_img(const FilterArgs& a, hipStream_t st);
void wino6_launch_filter_bwd(const FilterArgs& a, hipStream_t st);

}  // namespace lgd
