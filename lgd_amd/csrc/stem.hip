// K11: the frozen ResNet stem as ONE kernel  (include/lgd_hip.h: lgd_stem7_*)
//   out = max_pool2d(relu(conv2d(x, w, stride 2, padding 3) + shift[c]), 3, 2, 1),  x (N, 3, H, W), w (64, 3, 7, 7) with the FrozenBN scale folded in
//   [d2-memory: detectron2 BasicStem -- conv1 7x7/2 -> FrozenBN -> ReLU -> max_pool2d(3, 2, 1); frozen in every config of the reference
//    (MODEL.BACKBONE.FREEZE_AT = 2): forward only.  SURVEY.md appendix A; the student the distillator wraps, distillator.py:100-112]
// The library's best fp32 kernel for this convolution (a Winograd F(3x2) decomposition of the 7x7 / stride 2 filter) takes 830 us at 8 images of
// 800 x 1344 and writes a 550 MB map that the pooling pass reads back (170 us): 1.0 ms of the step for 40 GFLOP and 240 MB of algorithmic
// traffic.  Here the convolution is an implicit GEMM on v_mfma_f32_32x32x16_f16 from two-piece f16 operands (x 2^e = h + m, three of the four cross
// products, fp32 accumulate: the f16x2 form of csrc/h2.hip, error vs fp64 ~5e-7 of the output scale), and shift, ReLU and the 3x3 / 2 max-pool
// are applied to the accumulators: the conv output never exists in memory.
//   * k axis = (channel, filter row) pairs j = 7 c + ky, each with its 7 taps padded to 8: k = 8 j + kx, 22 pairs (21 + a zero one) = 11 k-steps
//     of 16.  A lane's B fragment (8 consecutive k) is then 8 CONSECUTIVE input pixels of one row of the tile's input patch -- two ds_read2_b32 from
//     the patch kept in LDS as f16 pieces -- and the eighth pixel meets a zero weight.
//   * workgroup = 8 x 15 pooled pixels of one image = conv rows 2 p0 - 1 .. 2 p0 + 15, conv columns 30 t - 1 .. 30 t + 30 (32 lanes; tiles step by 30
//     columns so that every pooling window lies inside one tile: 32 / 30 x 9 / 8 = 1.2x the convolution's flops, which the MFMA pipe has to spare).
//     Input patch 39 x 70 x 3 pixels, split once into f16 pieces (34 KB of LDS).
//   * wave = 32 output channels (the filter fragments of its channel block stay in 88 registers for the whole tile) x 9 conv rows, one row (one
//     32 x 32 MFMA block, 33 MFMAs) at a time; the pool's vertical maximum runs in registers over the rows as they are produced, the horizontal one
//     over the neighbouring lanes (ds_bpermute) when a pooled row is complete.
#include "winograd.h"   // h2_exponent / h2_pow2 / common.h

namespace lgd {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef LGD_STEM7_ABL
#define LGD_STEM7_ABL 0   // lab ablations (results are garbage): 1 no patch loads from memory, 2 no MFMAs, 3 no output stores, 4 no filter fragment loads
#endif
constexpr int kPR = 39, kPC = 72, kPCU = 70;      // patch rows, row stride (halves), columns used
constexpr int kKS = 11;                          // k-steps of 16
constexpr int kPoolR = 8, kPoolC = 15;           // pooled pixels per workgroup

struct Stem7Args {
    const float* x; const char* img; const float* w_inv; const unsigned* x_amax; const float* shift; float* out;
    unsigned* amax;   // optional: atomic max of the float bits of the pooled outputs (a word the caller zeroed): the magnitude tag of the stem's output
    int N, H, W, Ho, Wo, Hp, Wp;
};

__device__ __forceinline__ void split1_f16(float v, _Float16& h, _Float16& m) {
    asm volatile("" : "+v"(v));   // (keep the compiler from fusing the scale into the conversion: h must be the f16 of the fp32 product, csrc/winograd.h)
    h = (_Float16)v;
    m = (_Float16)(v - (float)h);
}

#ifndef LGD_STEM7_WAVES
#define LGD_STEM7_WAVES 2   // waves per SIMD the register allocator is held to (lab knob)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LGD_STEM7_WAVES))) void stem7_kernel(Stem7Args a) {
    __shared__ __attribute__((aligned(16))) _Float16 ph[3 * kPR * kPC], pm[3 * kPR * kPC];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), mb = w >> 1, half = w & 1;
    const int n_img = blockIdx.z, pr0 = blockIdx.y * kPoolR, pc0 = blockIdx.x * kPoolC;
    const int cy0 = 2 * pr0 - 1, cx0 = 2 * pc0 - 1;          // conv pixel of tile row / column 0
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // input pixel of patch row / column 0
    const int ex = h2_exponent(*a.x_amax, 0);
    const float sx = h2_pow2(ex), inv = h2_pow2(-ex) * a.w_inv[0];
    // the filter fragments of this wave's 32 channels: [k-step][piece][channel block][lane][8 halves]
    f16x8 ah[kKS], am[kKS];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
#if LGD_STEM7_ABL == 4
        for (int e = 0; e < 8; ++e) { ah[ks][e] = (_Float16)(lane + ks); am[ks][e] = (_Float16)(lane - ks); }
#else
        ah[ks] = *reinterpret_cast<const f16x8*>(a.img + ((ks * 2 + 0) * 2 + mb) * 1024 + lane * 16);
        am[ks] = *reinterpret_cast<const f16x8*>(a.img + ((ks * 2 + 1) * 2 + mb) * 1024 + lane * 16);
#endif
    }
    // the input patch, zero outside the image, as f16 pieces of x 2^ex
    const float* xi = a.x + (size_t)n_img * 3 * a.H * a.W;
    constexpr int kPatch = 3 * kPR * kPCU;
    for (int base = 0; base < kPatch; base += 256 * 8) {   // eight loads in flight per thread (one at a time the patch costs 32 memory latencies)
        float v[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + tid;
            const int c = i / (kPR * kPCU), rem = i - c * (kPR * kPCU), r = rem / kPCU, col = rem - r * kPCU;
            const int iy = iy0 + r, ix = ix0 + col;
            const bool in = i < kPatch && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#if LGD_STEM7_ABL == 1
            v[u] = in ? 1.f : 0.f;
#else
            v[u] = in ? xi[((size_t)c * a.H + iy) * a.W + ix] : 0.f;
#endif
            dst[u] = i < kPatch ? (c * kPR + r) * kPC + col : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            _Float16 h, m;
            split1_f16(v[u] * sx, h, m);
            if (dst[u] >= 0) { ph[dst[u]] = h; pm[dst[u]] = m; }
        }
    }
    const int g = lane >> 5, n = lane & 31;
    float sh[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) sh[e] = a.shift[32 * mb + (e & 3) + 8 * (e >> 2) + 4 * g];
    __syncthreads();
    const int cx = cx0 + n;
    const bool colok = cx >= 0 && cx < a.Wo;
    float cur[16], omax = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) cur[e] = 0.f;
#pragma unroll 1
    for (int r = 0; r < 9; ++r) {
        const int cr = 8 * half + r;                          // conv row of the tile
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int rowbase = (2 * cr) * kPC + 2 * n;           // halves: patch row 2 cr (+ ky), column 2 n (+ kx)
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            // pair j = 2 ks + g (the zero pair 21 re-reads pair 20: finite data against zero weights)
            const int j0 = 2 * ks, j1 = 2 * ks + 1 < 21 ? 2 * ks + 1 : 20;
            const int o0 = ((j0 / 7) * kPR + j0 % 7) * kPC, o1 = ((j1 / 7) * kPR + j1 % 7) * kPC;
            const int off = rowbase + (g ? o1 : o0);
            const uint32_t* qh = reinterpret_cast<const uint32_t*>(ph + off);
            const uint32_t* qm = reinterpret_cast<const uint32_t*>(pm + off);
            const lgd_u32x4 bhw = {qh[0], qh[1], qh[2], qh[3]}, bmw = {qm[0], qm[1], qm[2], qm[3]};
            const f16x8 bh = __builtin_bit_cast(f16x8, bhw), bm = __builtin_bit_cast(f16x8, bmw);
#if LGD_STEM7_ABL == 2
            acc[ks] += (float)bh[0] * (float)am[ks][1] + (float)bm[2] * (float)ah[ks][3];
#else
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[ks], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, acc, 0, 0, 0);
#endif
        }
        // conv pixels outside the conv output are the pool's padding: after the ReLU every window holds a value >= 0, so 0 stands in for -inf
        const int cy = cy0 + cr;
        const bool ok = colok && cy >= 0 && cy < a.Ho;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = ok ? fmaxf(fmaf(acc[e], inv, sh[e]), 0.f) : 0.f;
        if ((r & 1) == 0 && r > 0) {
            // conv row 2 i + 2 closes pooled row i: the horizontal maximum over lanes 2 q, 2 q + 1, 2 q + 2 lands on lane 2 q
            const int pr = pr0 + 4 * half + (r >> 1) - 1, pc = pc0 + (n >> 1);
            const bool st = (n & 1) == 0 && n <= 28 && pr < a.Hp && pc < a.Wp;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float t = fmaxf(cur[e], v[e]);
                const float t1 = __shfl_down(t, 1, 32), t2 = __shfl_down(t, 2, 32);
                const int m = 32 * mb + (e & 3) + 8 * (e >> 2) + 4 * g;
                const float o = fmaxf(t, fmaxf(t1, t2));
#if LGD_STEM7_ABL == 3
                if (st && t == 123456.f)
#else
                if (st)
#endif
                    a.out[(((size_t)n_img * 64 + m) * a.Hp + pr) * a.Wp + pc] = o;
                omax = fmaxf(omax, st ? o : 0.f);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[e] = (r & 1) == 0 ? v[e] : fmaxf(cur[e], v[e]);
    }
    if (a.amax) {
        __syncthreads();   // (the patch is dead: its first words serve as the four slots of the workgroup's maximum)
        block_max_bits(a.amax, wave_max(omax), reinterpret_cast<float*>(ph));
    }
}

// the filter as MFMA fragments: [k-step][piece][32-channel block][lane][8 halves] of w 2^e, e from the bound *amax of |w|; thread per fragment slot
__global__ __launch_bounds__(256) void stem7_image_kernel(const float* __restrict__ wt, const unsigned* __restrict__ amax, char* __restrict__ img,
                                                          float* __restrict__ inv_out) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int e = h2_exponent(*amax, 0);
    if (q == 0) inv_out[0] = h2_pow2(-e);
    if (q >= kKS * 2 * 64) return;
    const float s = h2_pow2(e);
    const int lane = q & 63, mbk = (q >> 6) & 1, ks = q >> 7;
    const int m = 32 * mbk + (lane & 31), j = 2 * ks + (lane >> 5);
    _Float16 h[8], mm[8];
#pragma unroll
    for (int kx = 0; kx < 8; ++kx) {
        const float v = (j < 21 && kx < 7) ? wt[((m * 3 + j / 7) * 7 + j % 7) * 7 + kx] * s : 0.f;
        split1_f16(v, h[kx], mm[kx]);
    }
    f16x8 hv, mv;
#pragma unroll
    for (int kx = 0; kx < 8; ++kx) { hv[kx] = h[kx]; mv[kx] = mm[kx]; }
    *reinterpret_cast<f16x8*>(img + ((ks * 2 + 0) * 2 + mbk) * 1024 + lane * 16) = hv;
    *reinterpret_cast<f16x8*>(img + ((ks * 2 + 1) * 2 + mbk) * 1024 + lane * 16) = mv;
}

}  // namespace
}  // namespace lgd

extern "C" {

size_t lgd_stem7_image_bytes(void) { return (size_t)lgd::kKS * 2 * 2 * 1024; }

int lgd_stem7_image(const float* w, const uint32_t* w_amax, void* image, float* w_inv, void* stream) {
    if (!w || !w_amax || !image || !w_inv || ((uintptr_t)image & 15)) return LGD_EINVAL;
    LGD_LAUNCH("stem7_image_kernel", lgd::stem7_image_kernel, dim3((lgd::kKS * 2 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, w_amax, (char*)image,
               w_inv);
    return lgd::check_launch();
}

int lgd_stem7_conv_pool(const float* x, const void* image, const float* w_inv, const uint32_t* x_amax, const float* shift, int N, int H, int W, float* out,
                        uint32_t* amax_out, void* stream) {
    if (!x || !image || !w_inv || !x_amax || !shift || !out || N < 1 || H < 1 || W < 1 || (long long)H * W > (1LL << 27)) return LGD_EINVAL;
    lgd::Stem7Args a;
    a.x = x; a.img = (const char*)image; a.w_inv = w_inv; a.x_amax = x_amax; a.shift = shift; a.out = out; a.amax = amax_out;
    a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1; a.Hp = (a.Ho - 1) / 2 + 1; a.Wp = (a.Wo - 1) / 2 + 1;
    if (N > 65535) return LGD_EINVAL;
    const dim3 grid((a.Wp + lgd::kPoolC - 1) / lgd::kPoolC, (a.Hp + lgd::kPoolR - 1) / lgd::kPoolR, N);
    if (grid.y > 65535) return LGD_EINVAL;
    LGD_LAUNCH("stem7_kernel", lgd::stem7_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
