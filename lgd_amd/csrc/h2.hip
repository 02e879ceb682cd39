// K10 (round 5): the Winograd channel products on the f16 MFMA pipe from TWO-piece split operands that are already split in HBM
// (include/lgd_hip.h: lgd_h2_*).  Same arithmetic contract as K9 (csrc/gemm3.hip): an fp32-class product of fp32 data --
//   [the arithmetic of nn.Conv2d(C, C', 3, padding=1): dynamic_teacher.py:57,61,67-73,145,280; sequential_convs.py:10-12; the head towers,
//    distillator.py:107-109] --
// but the operands arrive as   x 2^e = h + m,  h = f16(x 2^e),  m = f16(x 2^e - h)   (round to nearest: |x 2^e - h - m| <= 2^-23 |x 2^e|)
// and the product keeps   a b ~ ah bh + ah bm + am bh   (dropped: am bm <= 2^-22 |a||b|), accumulated in fp32 inside
// v_mfma_f32_32x32x16_f16: THREE MFMAs per k-step instead of gemm3's six, and 4 bytes per element in HBM -- the same as fp32 -- so the
// Winograd data transforms write V and dM split (csrc/winograd6.hip, H2 variants: their traffic does not change) and the product kernels
// split nothing: both operands come by LDS-DMA, no registers and no VALU on the way.  Measured against an fp64 product of the same fp32
// operands (tools/h2_lab.py, profiles/r05_experiment_h2_lab_v1.log): forward 5.8e-7 of the output scale (library fp32 GEMM: 1.0e-6),
// weight gradient 7.6e-7 (library: 4.1e-6); through the F(6x6,3x3) transforms the convolution error equals the fp32 pipeline's
// (tools/lab/split_numerics.py).  f16 has 5 exponent bits: every operand carries ONE power-of-two scale per batch (frequency), chosen
// from a guaranteed bound of its magnitude (the maps' max |x| times the abs row sums of the transform matrices), so nothing overflows
// for any input; an element more than 2^18 below the bound has its remainder m in the f16 SUBNORMAL range (spacing 2^-24), so its error becomes
// absolute: <= 2^-25 in scaled units = between 2^-40 and 2^-39 of the bound (the bound lands in [2^14, 2^15)) instead of 2^-23 |x| -- graceful, and
// pinned by tests/test_h2_gpu.py::test_h2_within_plane_dynamic_range (the MFMA pipe takes f16 subnormals as they are; a tag 16x too wide costs
// exactly 16x in that regime and nothing near the bound).
// The products are rescaled by 2^-(ea + eb) on the way out.
//
// Operand formats
//   split rows (V, dM):  row r of batch b at  base + r * rs + b * sb  (bytes);  tile t inside the row: h2_piece_off(t, piece) (winograd.h:
//                        blocks of 32 tiles, 64 bytes of h then 64 bytes of m)
//   image (U, U^T):      [batch][k-step of 16][piece][32-row block][lane = (k % 16 / 8) * 32 + row % 32][8 f16]   (wino6_filter_img_kernel<true>)
// Kernels
//   h2_fwd_kernel:  C[b] (M x N) = A[b] (image, M x K) . B[b] (split rows = k, n contiguous)            M = U V,  dV = U^T dM
//       256 x 128 x 16 tile, 4 waves (128 x 64 each), three LDS buffers of 24 KB, two workgroups per CU.  B is k-strided / n-contiguous:
//       its LDS image is [piece][k][256 B] with every row rotated by 64 B x (k % 4), and the fragments are taken with ds_read_b64_tr_b16
//       (conflict-free: a half-wave's 4 rows x 2 column groups tile one 256-byte bank row).
//   h2_dw_kernel:   P[s][b] (M x N) = sum over the k-range of split s of A[b][m][k] B[b][n][k]            dU = dM V^T
//       256 x 256 x 16 tile, 8 waves, four LDS buffers of 32 KB (one workgroup per CU): every operand byte is read from HBM once; split-K
//       over s with a fixed-order reduction (h2_reduce_kernel): bit-reproducible.
// The LDS-DMA is issued from inline asm: hipcc's wait insertion drains vmcnt(0) in front of every LDS read and every barrier that follows
// a builtin LDS-DMA in program order, which would serialise the pipeline.  The kernels count their own waits: the memory pipe returns
// loads in order, each k-step issues exactly DPW DMA instructions per wave and nothing else touches vmcnt inside the k-loop
// (tests/test_abi.py::test_h2_kloop_has_only_counted_waits disassembles the object and checks exactly that).
#include <stdlib.h>

#include <type_traits>

#include "winograd.h"

#ifndef LGD_H2_ABL
#define LGD_H2_ABL 0   // lab ablations of h2_fwd_kernel (tools/h2_ablate.sh; results are garbage): 1 no C stores, 2 no DMA of B, 3 no DMA of the image,
#endif                 // 4 no MFMAs, 5 no DMA at all and no stores (fragment reads, MFMAs and barriers only)

namespace lgd {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long bytes_left) {
    const uint32_t n = bytes_left <= 0 ? 0u : bytes_left > 0xffffffffL ? 0xffffffffu : (uint32_t)bytes_left;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
// 16 bytes per lane, global -> LDS (lane-linear: LDS byte address lds_addr + 16 lane).  The statement WRITES m0 (the LDS base of the DMA) and says
// so: without the clobber the compiler may keep a value of its own in m0 across it (its LDS-DMA builtins, v_movrel, s_sendmsg, ds_gws / readlane
// forms all set m0 up ahead of use) or move such a set-up across it -- ADVICE r5; tests/test_abi.py::test_h2_dma_statements_own_m0 checks the object.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // "clobber list contains reserved registers: m0": reserved = never allocated, which is why it has to be said
__device__ __forceinline__ void glds16(const __amdgpu_buffer_rsrc_t rs, uint32_t lds_addr, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }

// ------------------------------------------------------------------------------------------------------------------ forward / dx product
struct FwdP {
    const char* Aimg; long a_sb; int rbp;            // image; a_sb bytes per batch
    const char* B; long b_sb, b_ld, b_bytes;         // split rows: batch stride, k-row stride, extent from B (bytes)
    float* C; long c_sb, c_ld;                       // floats
    const float* a_inv; const float* b_inv; int b_inv_stride;   // 2^-e per batch (stride 0: one scale for all batches)
    unsigned* amax_out;                              // AMAX kernels: per batch max |C| (float bits, atomicMax; pre-zeroed)
    int nb, M, N, K, mt, nt;
};

template <int BM, bool AMAX>
__global__ __launch_bounds__(256) void h2_fwd_kernel(const FwdP p) {
    constexpr int RB = BM / 32, MI = BM / 64, BN = 128;
    constexpr int A_BYTES = 2 * RB * 1024, B_BYTES = 8192, BUF = A_BYTES + B_BYTES, ST = 3;
    constexpr int ACH = 2 * RB / 4, DPW = ACH + 2;   // LDS-DMA instructions per wave and k-step
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 1, wn = w & 1;
    // workgroup -> (batch, n-tile, m-tile): consecutive ids go round-robin to the 8 XCDs; XCD x takes the batches b = x (mod 8), so that its
    // L2 holds the images of the one or two batches it is working on (as in gemm3)
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int per_b = p.nt * p.mt;
    const int b = (j / per_b) * 8 + xcd;
    const int r = j % per_b;
    if (b >= p.nb) return;
    const int tn = r / p.mt, sub = r % p.mt;
    const int m0 = sub * BM, n0 = tn * BN, rb0 = sub * RB;
    const int ksteps = p.K / 16;
    const char* Ai = p.Aimg + (long)b * p.a_sb;
    const char* Bb = p.B + (long)b * p.b_sb;
    uint32_t aoff[ACH];
#pragma unroll
    for (int c = 0; c < ACH; ++c) {
        const int ch = w * ACH + c, pc = ch / RB, rbl = ch % RB;
        int rb = rb0 + rbl;
        rb = rb < p.rbp ? rb : p.rbp - 1;   // row blocks past the image repeat its last one: their products are rows >= M
        aoff[c] = (uint32_t)((pc * p.rbp + rb) * 1024 + lane * 16);
    }
    const long astep = (long)2 * p.rbp * 1024;
    // B: instruction jb = 2 w + jj moves piece jb >> 2, rows 4 (jb & 3) .. + 3; lane -> (row = lane >> 4, LDS chunk q = lane & 15), which holds
    // the row's 16-byte chunk (q - 4 row) & 15 (chunk = 8 columns): the rotation by 64 bytes per row that makes the transposing reads
    // conflict-free.  Columns >= N read other rows' data or, past the buffer, zeros: they only reach elements of C that are never stored.
    uint32_t boff[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int jb = 2 * w + jj, pc = jb >> 2, rg = jb & 3, rl = lane >> 4, q = lane & 15;
        const int sc = (q - 4 * rl) & 15, rr = 4 * rg + rl;
        boff[jj] = (uint32_t)((long)rr * p.b_ld + h2_piece_off(n0 + 8 * sc, pc));
    }
    const long bstep = 16 * p.b_ld;
    const long b_left0 = p.b_bytes - (long)b * p.b_sb;
    auto dma = [&](int ks, int buf) {
        char* dst = lds + buf * BUF;
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ai + ks * astep, 0x7fffffff);
#if LGD_H2_ABL != 3 && LGD_H2_ABL != 5
#pragma unroll
        for (int c = 0; c < ACH; ++c) glds16(ra, lds_addr_of(dst + (w * ACH + c) * 1024), aoff[c]);
#endif
        const __amdgpu_buffer_rsrc_t rb_ = make_rsrc(Bb + ks * bstep, b_left0 - ks * bstep);
#if LGD_H2_ABL != 2 && LGD_H2_ABL != 5
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) glds16(rb_, lds_addr_of(dst + A_BYTES + (2 * w + jj) * 1024), boff[jj]);
#endif
    };
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) asm volatile("" : "+a"(acc[i][jn]));
    // transposing reads of B (ds_read_b64_tr_b16: within a 16-lane group lane i supplies the address of row i >> 2, 4 columns at (i & 3) 4, and
    // receives column i's 4 rows): lane -> group cg = (lane >> 4) & 1 (columns cg 16 ..), k-group g = lane >> 5 (k = 8 g ..), i = lane & 15:
    // source row 8 g + 4 h + (i >> 2); position inside the rotated 256-byte row: (2 col + 64 (i >> 2)) & 255
    const int g = lane >> 5, cg = (lane >> 4) & 1, i16 = lane & 15;
    int btr[2];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int col = wn * 64 + jn * 32 + cg * 16 + (i16 & 3) * 4;
        btr[jn] = A_BYTES + (8 * g + (i16 >> 2)) * 256 + ((2 * col + 64 * (i16 >> 2)) & 255);
    }
    const int slot = lane * 16;
    // Pipeline: three buffers; the k-step's barrier sits at the END of the iteration behind a counted wait (the youngest group of DPW DMAs
    // may still be in flight), the fragment reads open the next one and the DMA for k-step ks + 2 is issued right behind them, into the buffer
    // everybody left before the previous barrier.
    dma(0, 0);
    dma(ksteps > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const char* cur = lds + (ks % ST) * BUF;
        f16x8 fa[2][MI], fb[2][2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
#if LGD_H2_ABL == 6   // lab: plain 8-byte LDS reads at the same addresses instead of the transposing ones (garbage fragments, same traffic)
                const fp16x4 lo = *(const fp16x4*)(cur + btr[jn] + pc * 4096);
                const fp16x4 hi = *(const fp16x4*)(cur + btr[jn] + pc * 4096 + 1024);
#else
                const fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(cur + btr[jn] + pc * 4096));
                const fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(cur + btr[jn] + pc * 4096 + 1024));
#endif
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                fb[pc][jn] = __builtin_bit_cast(f16x8, (u32x4){l2[0], l2[1], h2[0], h2[1]});
            }
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[pc][i] = *reinterpret_cast<const f16x8*>(cur + pc * (RB * 1024) + (wm * MI + i) * 1024 + slot);
        {
            const int nx = ks + 2 < ksteps ? ks + 2 : ksteps - 1;   // (the last ones re-read a valid k-step into a dead buffer: the count stays DPW)
            dma(nx, (ks + 2) % ST);
        }
        // smallest terms first
#if LGD_H2_ABL == 4   // (lab: the fragments stay live through one cheap use)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) acc[i][jn][0] += (float)fa[1][i][0] + (float)fb[0][jn][0] + (float)fa[0][i][1] + (float)fb[1][jn][1];
#else
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i], fb[0][jn], acc[i][jn], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[1][jn], acc[i][jn], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][jn], acc[i][jn], 0, 0, 0);
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5); buffer stores, one running offset
    const float inv = p.a_inv[b] * p.b_inv[b * p.b_inv_stride];
    const int rr = lane & 31;
    const int wrow = m0 + wm * (BM / 2), wcol = n0 + wn * 64;
    const int mw = wrow + 4 * g, nw = wcol + rr;
    const int ld = (int)p.c_ld;
    const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
    const __amdgpu_buffer_rsrc_t cs = make_rsrc(p.C + (long)b * p.c_sb + (long)wrow * ld + wcol, 0x7fffffff);
    const int c1 = ld * 4, c5 = ld * 20, mrem = p.M - mw;
    const bool colok[2] = {nw < p.N, nw + 32 < p.N};
    uint32_t amax = 0u;
    auto epi = [&](auto HF) {
        constexpr bool hf = decltype(HF)::value;
        int cbase = (4 * g * ld + rr) * 4;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                asm volatile("" : "+a"(acc[i][jn])::"memory");
                int co = cbase;
                uint32_t bm = 0u;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int dm = i * 32 + (e & 3) + 8 * (e >> 2);
                    const float v = acc[i][jn][e] * inv;
#if LGD_H2_ABL == 1 || LGD_H2_ABL == 5   // lab: no C stores (a never-true condition keeps the accumulators alive)
                    const bool ok = v == 123456.789f;
#else
                    const bool ok = hf || (dm < mrem && colok[jn]);
#endif
                    if (ok) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), cs, co + jn * 128, 0, 0);
                    if constexpr (AMAX) { const uint32_t ab = __builtin_bit_cast(uint32_t, v) & 0x7fffffffu; bm = max(bm, ok ? ab : 0u); }
                    co += (e & 3) == 3 ? c5 : c1;
                    asm volatile("" : "+v"(co));
                }
                if constexpr (AMAX) { amax = max(amax, bm); asm volatile("" : "+v"(amax)); }
                if (jn == 1) cbase = co;
            }
        }
    };
    if (full) epi(std::true_type()); else epi(std::false_type());
    if constexpr (AMAX) {   // (non-negative floats order like their bit patterns; NaN bits sort above everything and propagate)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, o));
        if (lane == 0) atomic_max_bits(p.amax_out + b, amax);
    }
}

// ------------------------------------------------------------------------------------------------------------------ weight-gradient product
struct DwP {
    const char* A; long a_rs, a_sb, a_bytes;       // split rows (m): row stride, batch stride, extent (bytes)
    const char* B; long b_rs, b_sb, b_bytes;       // split rows (n)
    float* P;                                      // partials [S][nb][M][N] (S == 1: the result)
    const float* a_inv; const float* b_inv; int a_inv_stride, b_inv_stride;
    int nb, M, N, nstage, S, per, mt, nt;          // nstage: 16-tile k-stages in total; per: stages per split
};

__global__ __launch_bounds__(512) void h2_dw_kernel(const DwP p) {
    constexpr int KC = 4, ST = 4;                                // 16-byte chunks per row and stage (16 tiles x 2 pieces), LDS buffers
    constexpr int ROWB = KC * 16, OPB = 256 * ROWB, BUF = 2 * OPB;
    constexpr int RPI = 64 / KC, IPW = (256 / RPI) / 8;          // rows per DMA instruction, instructions per wave and operand
    constexpr int DPW = 2 * IPW;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 2, wn = w & 3;
    int id = blockIdx.x;
    const int tn = id % p.nt; id /= p.nt;
    const int tm = id % p.mt; id /= p.mt;
    const int s = id % p.S;
    const int b = id / p.S;
    const int m0 = tm * 256, n0 = tn * 256;
    const int st0 = s * p.per, st1 = min(st0 + p.per, p.nstage), nst = st1 - st0;
    float* P = p.P + ((long)s * p.nb + b) * p.M * p.N;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;
    const int rib = lane & 31, g = lane >> 5;
    if (nst > 0) {   // (wave-uniform; an empty split writes zeros)
    // DMA: instruction jj of wave w covers rows (w IPW + jj) RPI .. of the tile; lane -> (row = lane / KC, LDS chunk q = lane % KC) which holds the
    // row's chunk q ^ swz(row), chunk = (piece, octet) = (c >> 1, c & 1): the fragment reads of a ds_read_b128 lane group then touch 16
    // distinct 16-byte slots of a 256-byte bank row
    const int rl = lane / KC, q = lane % KC;
    uint32_t offA[IPW], offB[IPW];
#pragma unroll
    for (int jj = 0; jj < IPW; ++jj) {
        const int row = jj * RPI + rl;   // relative to the wave's first row (a multiple of 32: the swizzle sees the same bits)
        const int sc = q ^ ((row >> 2) & 3);
        const int po = (sc >> 1) * (2 * kH2Block) + (sc & 1) * 16;
        offA[jj] = (uint32_t)((long)row * p.a_rs + po);
        offB[jj] = (uint32_t)((long)row * p.b_rs + po);
    }
    const long a0 = (long)b * p.a_sb + (long)(m0 + w * IPW * RPI) * p.a_rs, b0 = (long)b * p.b_sb + (long)(n0 + w * IPW * RPI) * p.b_rs;
    auto dma = [&](int stg, int buf) {
        char* dst = lds + buf * BUF;
        const long ko = h2_piece_off((long long)stg * 16, 0);
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + a0 + ko, p.a_bytes - a0 - ko);
#pragma unroll
        for (int jj = 0; jj < IPW; ++jj) glds16(ra, lds_addr_of(dst + (w * IPW + jj) * 1024), offA[jj]);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B + b0 + ko, p.b_bytes - b0 - ko);
#pragma unroll
        for (int jj = 0; jj < IPW; ++jj) glds16(rb, lds_addr_of(dst + OPB + (w * IPW + jj) * 1024), offB[jj]);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) asm volatile("" : "+a"(acc[i][jn]));
    // fragment reads: lane -> (row in block rib = lane & 31, k-group g = lane >> 5); chunk (piece 2 + g) ^ swz(rib)
    const int sw = (rib >> 2) & 3;
    const int abase = (wm * 128 + rib) * ROWB, bbase = OPB + (wn * 64 + rib) * ROWB;
    int xo[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) xo[pc] = ((pc * 2 + g) ^ sw) * 16;
#pragma unroll
    for (int i = 0; i < ST - 1; ++i) dma(st0 + (i < nst ? i : nst - 1), i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * DPW) : "memory");
    __syncthreads();
    for (int it = 0; it < nst; ++it) {
        const char* cur = lds + (it % ST) * BUF;
        f16x8 fa[2][4], fb[2][2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[pc][i] = *reinterpret_cast<const f16x8*>(cur + abase + i * 32 * ROWB + xo[pc]);
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) fb[pc][jn] = *reinterpret_cast<const f16x8*>(cur + bbase + jn * 32 * ROWB + xo[pc]);
        }
        {
            const int nx = it + ST - 1 < nst ? it + ST - 1 : nst - 1;
            dma(st0 + nx, (it + ST - 1) % ST);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i], fb[0][jn], acc[i][jn], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[1][jn], acc[i][jn], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][jn], acc[i][jn], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * DPW) : "memory");
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const float inv = p.a_inv[b * p.a_inv_stride] * p.b_inv[b * p.b_inv_stride];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            asm volatile("" : "+a"(acc[i][jn])::"memory");
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * g, n = n0 + wn * 64 + jn * 32 + rib;
                if (m < p.M && n < p.N) P[(long)m * p.N + n] = acc[i][jn][e] * inv;
            }
        }
}

__global__ void h2_reduce_kernel(const float* __restrict__ P, float* __restrict__ out, long n4, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 a = reinterpret_cast<const float4*>(P)[i];
    for (int s = 1; s < S; ++s) {
        const float4 v = reinterpret_cast<const float4*>(P)[(long)s * n4 + i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
}

// ------------------------------------------------------------------------------------------------------------------ 1x1 weight gradient
// P[s] (M x N) = sum over the (image, pixel) range of split s of A[n][m][px] B[n][k][px]:  dW of a pointwise convolution, A = dz (N, M, HW),
// B = x (N, Nn, HW) fp32 NCHW maps.  Both operands are activations that exist only as fp32: they are scaled (2^ea, 2^eb from their bounds) and
// split into f16 pairs in REGISTERS -- 3 VALU per element, against 24 MFMAs per 16 pixels of a 256 x 256 tile: a quarter of the MFMA time, where
// the bf16x3 form of both operands (11 per pair, 48 MFMAs) was issue-bound at the library's speed.  256 x 256 x 32 tile,
// 8 waves (128 x 64 each), register double buffering (the loads of stage s + 1 fly under the MFMAs of stage s), two 64 KB LDS buffers, rows of
// 128 B = 8 chunks [sub-step][piece][k-group] swizzled by (row >> 1) & 7 like the DMA-fed form; split-K over the flat (image, 32-pixel stage) index,
// partials summed (and scaled by a frozen per-row factor) by lgd_sum_batch_scale.  HW % 4 == 0.
struct PwDwP {
    const float* A; const float* B;      // (n, M, HW), (n, N, HW) contiguous
    float* P;                            // partials [S][M][N]
    const unsigned* a_amax; const unsigned* b_amax;
    int nimg, M, N, HW, spi, nstage, S, per, mt, nt;   // spi: 32-pixel stages per image
};

__global__ __launch_bounds__(512) void h2_pwdw_kernel(const PwDwP p) {
    constexpr int ROWB = 128, OPB = 256 * ROWB, BUF = 2 * OPB;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 2, wn = w & 3;
    int id = blockIdx.x;
    const int tn = id % p.nt; id /= p.nt;
    const int tm = id % p.mt; id /= p.mt;
    const int s = id;
    const int m0 = tm * 256, n0 = tn * 256;
    const int st0 = s * p.per, st1 = min(st0 + p.per, p.nstage), nst = st1 - st0;
    const int ea = h2_exponent(*p.a_amax, 0), eb = h2_exponent(*p.b_amax, 0);
    const float sa = h2_pow2(ea), sb = h2_pow2(eb);
    // loads: thread -> (row = t >> 3 (+ 64 i), 16-byte chunk q = t & 7 of the row's 32 pixels); a row's 128 bytes = 8 consecutive lanes
    const int q = t & 7, r0 = t >> 3;
    const float* pa[4]; const float* pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = min(m0 + r0 + 64 * i, p.M - 1), rb = min(n0 + r0 + 64 * i, p.N - 1);   // rows past the operand repeat its last one: never stored
        pa[i] = p.A + (size_t)ra * p.HW + 4 * q;
        pb[i] = p.B + (size_t)rb * p.HW + 4 * q;
    }
    const size_t a_img = (size_t)p.M * p.HW, b_img = (size_t)p.N * p.HW;
    float4 va[4], vb[4];
    auto load = [&](int stg) {
        const int n = stg / p.spi, ps = stg - n * p.spi;
        const int px = 32 * ps + 4 * q;
        const bool ok = px < p.HW;       // (HW % 4 == 0: a chunk lies inside the plane or beyond it)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            va[i] = ok ? ldg_stream4(pa[i] + n * a_img + 32 * ps) : make_float4(0.f, 0.f, 0.f, 0.f);
            vb[i] = ok ? ldg_stream4(pb[i] + n * b_img + 32 * ps) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // LDS: chunk c = sub * 4 + piece * 2 + g holds 8 consecutive pixels (16 * sub + 8 * g ..) of one piece; this thread's 4 pixels are half q & 1 of
    // chunk (sub = q >> 2, g = (q >> 1) & 1)
    const int csub = q >> 2, cg = (q >> 1) & 1, half = q & 1;
    auto store = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 64 * i, sw = (row >> 1) & 7;
            uint32_t h0, m0_, h1, m1_;
            {
                typedef _Float16 hx2 __attribute__((ext_vector_type(2)));
                typedef float fx2 __attribute__((ext_vector_type(2)));
                auto sp = [](float t0, float t1, uint32_t& h, uint32_t& m) {
                    const hx2 hh = __builtin_convertvector((fx2){t0, t1}, hx2);
                    const hx2 mm = __builtin_convertvector((fx2){t0 - (float)hh[0], t1 - (float)hh[1]}, hx2);
                    h = __builtin_bit_cast(uint32_t, hh); m = __builtin_bit_cast(uint32_t, mm);
                };
                sp(va[i].x * sa, va[i].y * sa, h0, m0_); sp(va[i].z * sa, va[i].w * sa, h1, m1_);
                char* d = buf + row * ROWB + half * 8;
                *reinterpret_cast<u32x2*>(d + (((csub * 4 + cg) ^ sw) << 4)) = (u32x2){h0, h1};
                *reinterpret_cast<u32x2*>(d + (((csub * 4 + 2 + cg) ^ sw) << 4)) = (u32x2){m0_, m1_};
                sp(vb[i].x * sb, vb[i].y * sb, h0, m0_); sp(vb[i].z * sb, vb[i].w * sb, h1, m1_);
                d += OPB;
                *reinterpret_cast<u32x2*>(d + (((csub * 4 + cg) ^ sw) << 4)) = (u32x2){h0, h1};
                *reinterpret_cast<u32x2*>(d + (((csub * 4 + 2 + cg) ^ sw) << 4)) = (u32x2){m0_, m1_};
            }
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;
    const int rib = lane & 31, g = lane >> 5;
    // a wave whose 128 rows (64 columns) lie beyond M (N) multiplies nothing: the other wave of its SIMD gets the pipe (C' = 128 layers)
    const bool live = m0 + wm * 128 < p.M && n0 + wn * 64 < p.N;
    if (nst > 0) {
        const int sw = (rib >> 1) & 7;
        const int abase = (wm * 128 + rib) * ROWB, bbase = OPB + (wn * 64 + rib) * ROWB;
        load(st0);
        store(lds);
        __syncthreads();
        for (int it = 0; it < nst; ++it) {
            const char* cur = lds + (it & 1) * BUF;
            if (it + 1 < nst) load(st0 + it + 1);
            if (live) {
#pragma unroll
                for (int sb_ = 0; sb_ < 2; ++sb_) {
                    f16x8 fa[2][4], fb[2][2];
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const int xo = ((sb_ * 4 + pc * 2 + g) ^ sw) << 4;
#pragma unroll
                        for (int i = 0; i < 4; ++i) fa[pc][i] = *reinterpret_cast<const f16x8*>(cur + abase + i * 32 * ROWB + xo);
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) fb[pc][jn] = *reinterpret_cast<const f16x8*>(cur + bbase + jn * 32 * ROWB + xo);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i], fb[0][jn], acc[i][jn], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[1][jn], acc[i][jn], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][jn], acc[i][jn], 0, 0, 0);
                }
            }
            if (it + 1 < nst) store(lds + ((it + 1) & 1) * BUF);   // (the other buffer: everybody left it before the previous barrier)
            __syncthreads();
        }
    }
    const float inv = h2_pow2(-(ea + eb));
    float* P = p.P + (size_t)s * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * g, n = n0 + wn * 64 + jn * 32 + rib;
                if (m < p.M && n < p.N) P[(size_t)m * p.N + n] = acc[i][jn][e] * inv;
            }
}

// ------------------------------------------------------------------------------------------------------------------ bounds of the operands
// max |act(x)| over a list of NCHW maps (the pyramid levels), act = identity | relu(x + bias[c]) | relu(x * scale + shift) per (level, image,
// channel): the bound the input transform's f16 scale is derived from.  Workgroup per 4096-element chunk of a plane.
constexpr int kAmaxChunk = 16384;   // elements of a plane per workgroup
struct AmaxArgs {
    const float* maps[LGD_MAX_LEVELS];
    unsigned blk_off[LGD_MAX_LEVELS + 1];
    int hw[LGD_MAX_LEVELS], cpp[LGD_MAX_LEVELS], chunk[LGD_MAX_LEVELS];   // plane size, chunks per plane, elements per chunk (balanced, a multiple of 4)
    const float* bias; const float* affine;
    unsigned* out;
    int L, N, C;
};

__global__ __launch_bounds__(256) void h2_amax_maps_kernel(AmaxArgs a) {
    __shared__ float slots[4];
    int l = 0;
#pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i)
        if (i < a.L && blockIdx.x >= a.blk_off[i]) l = i;
    l = __builtin_amdgcn_readfirstlane(l);
    const unsigned rel = blockIdx.x - a.blk_off[l];
    const int cpp = a.cpp[l], hw = a.hw[l];
    const unsigned plane = rel / cpp;           // n * C + c
    const int ch = rel - plane * cpp;
    const float* x = a.maps[l] + (size_t)plane * hw;
    float s = 1.f, sh = 0.f;
    const bool pre = a.bias || a.affine;
    if (a.affine) { const float2 v = reinterpret_cast<const float2*>(a.affine)[(size_t)l * a.N * a.C + plane]; s = v.x; sh = v.y; }
    else if (a.bias) sh = a.bias[plane % a.C];
    float am = 0.f;
    const int e0 = ch * a.chunk[l], e1 = min(hw, e0 + a.chunk[l]);   // (balanced: a 16,800-pixel p3 plane is two chunks of 8,400, not 16,384 + 416)
    if ((hw & 3) == 0) {
        for (int base = e0; base < e1; base += 4096) {   // four 16-byte loads per thread in flight
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = base + (k * 256 + threadIdx.x) * 4;
                v[k] = e < e1 ? ldg_stream4(x + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = base + (k * 256 + threadIdx.x) * 4;
                if (e >= e1) continue;
                if (pre) am = fmaxf(am, fmaxf(fmaxf(fmaf(v[k].x, s, sh), fmaf(v[k].y, s, sh)), fmaxf(fmaf(v[k].z, s, sh), fmaf(v[k].w, s, sh))));
                else am = fmaxf(am, fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w))));
            }
        }
    } else {
        for (int e = e0 + threadIdx.x; e < e1; e += 256) am = fmaxf(am, pre ? fmaf(x[e], s, sh) : fabsf(x[e]));
    }
    am = wave_max(am);   // (pre: relu(z) >= 0 and am starts at 0: max(relu(z)) = max(0, max z))
    block_max_bits(a.out, am, slots);
}

// max |w[co][..] * scale[co]| over K filter tensors (rows = output channels of `row` elements each)
struct AmaxFilterArgs { const float* w[8]; const float* scale[8]; int rows[8]; unsigned row_off[9]; unsigned* out; int K, row; };
__global__ __launch_bounds__(256) void h2_amax_filter_kernel(AmaxFilterArgs a) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < a.K && blockIdx.x >= a.row_off[i]) k = i;
    const int co = blockIdx.x - a.row_off[k];
    const float* w = a.w[k] + (size_t)co * a.row;
    const float s = a.scale[k] ? fabsf(a.scale[k][co]) : 1.f;
    float am = 0.f;
    for (int e = threadIdx.x; e < a.row; e += 256) am = fmaxf(am, fabsf(w[e]));
    __shared__ float slots[4];
    am = wave_max(am) * s;
    block_max_bits(a.out, am, slots);
}

// bound of |dx| = |adjoint input transform of dV| from the per-frequency maxima of dV (the dx product's AMAX epilogue), for the fused backward
// link (wino6_in_t_kernel<true, true>): window pixel (r, c) of ONE tile is at most W[r][c] = sum_ab |B^T[a][r]| |B^T[b][c]| amax[a][b]; a pixel of
// the tile's own block (rows / columns 1..6) also receives row / column 7 of the upper / left tile when r / c = 1 and row / column 0 of the
// lower / right tile when r / c = 6
__global__ __launch_bounds__(64) void h2_link_bound_kernel(const unsigned* __restrict__ amax64, unsigned* __restrict__ out) {
    constexpr float BT[8][8] = {{1.f, 0.f, 5.25f, 0.f, 5.25f, 0.f, 1.f, 0.f},      {0.f, 1.f, 1.f, 4.25f, 4.25f, 1.f, 1.f, 0.f},
                                {0.f, 1.f, 1.f, 4.25f, 4.25f, 1.f, 1.f, 0.f},      {0.f, 0.5f, 0.25f, 2.5f, 1.25f, 2.f, 1.f, 0.f},
                                {0.f, 0.5f, 0.25f, 2.5f, 1.25f, 2.f, 1.f, 0.f},    {0.f, 2.f, 4.f, 2.5f, 5.f, 0.5f, 1.f, 0.f},
                                {0.f, 2.f, 4.f, 2.5f, 5.f, 0.5f, 1.f, 0.f},        {0.f, 1.f, 0.f, 5.25f, 0.f, 5.25f, 0.f, 1.f}};   // |B^T|
    __shared__ float am[64], W[8][8];
    const int t = threadIdx.x, r = t >> 3, c = t & 7;
    am[t] = __builtin_bit_cast(float, amax64[t]);
    __syncthreads();
    float wsum = 0.f;
#pragma unroll
    for (int a_ = 0; a_ < 8; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 8; ++b_) wsum = fmaf(BT[a_][r] * BT[b_][c], am[a_ * 8 + b_], wsum);
    W[r][c] = wsum;
    __syncthreads();
    float tot = 0.f;
    if (r >= 1 && r <= 6 && c >= 1 && c <= 6) {
        const int r2 = r == 1 ? 7 : (r == 6 ? 0 : -1), c2 = c == 1 ? 7 : (c == 6 ? 0 : -1);
        tot = W[r][c];
        if (r2 >= 0) tot += W[r2][c];
        if (c2 >= 0) tot += W[r][c2];
        if (r2 >= 0 && c2 >= 0) tot += W[r2][c2];
    }
    tot = wave_max(tot) * 1.0001f;   // (rounding of the sums above)
    if (t == 0) out[0] = __builtin_bit_cast(unsigned, tot);
}

// per-channel sum over the tiles of ONE frequency plane of a split buffer [C][64][T] (rows of T / 32 blocks: 32 f16 h, 32 f16 m), times the plane's
// inverse scale: the bias gradient of a convolution is the sum of dM at the frequency of the interpolation point 1 (a tile's gradient sum) --
// one launch instead of the convert / reduce / scale passes of the tensor library
__global__ __launch_bounds__(256) void h2_plane_sums_kernel(const char* __restrict__ buf, long long ch_bytes, int T, const float* __restrict__ inv,
                                                            float* __restrict__ out) {
    __shared__ float slots[4];
    typedef _Float16 hx8 __attribute__((ext_vector_type(8)));
    const char* row = buf + (long long)blockIdx.x * ch_bytes;
    float s = 0.f;
    for (int i = threadIdx.x; i < T / 4; i += 256) {   // 16 bytes = 8 halves (h or m alike: both are summed)
        const hx8 v = *reinterpret_cast<const hx8*>(row + (size_t)i * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[e];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ((slots[0] + slots[1]) + (slots[2] + slots[3])) * inv[0];
}

struct WordsArgs { const unsigned* w[16]; int n; unsigned* out; };
__global__ __launch_bounds__(64) void h2_words_max_kernel(WordsArgs a) {
    unsigned v = threadIdx.x < (unsigned)a.n ? *a.w[threadIdx.x] : 0u;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    if (threadIdx.x == 0) *a.out = max(*a.out, v);
}

constexpr int kFwdLds256 = 3 * (2 * 8 * 1024 + 8192), kFwdLds128 = 3 * (2 * 4 * 1024 + 8192), kDwLds = 4 * 2 * 256 * 64, kPwDwLds = 2 * 2 * 256 * 128;

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: set once per (kernel, device)
int ensure_attrs() {
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LGD_ELAUNCH;
    if (done[dev]) return LGD_OK;
    const void* f256[2] = {(const void*)h2_fwd_kernel<256, false>, (const void*)h2_fwd_kernel<256, true>};
    const void* f128[2] = {(const void*)h2_fwd_kernel<128, false>, (const void*)h2_fwd_kernel<128, true>};
    for (const void* f : f256)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds256) != hipSuccess) return LGD_ELAUNCH;
    for (const void* f : f128)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds128) != hipSuccess) return LGD_ELAUNCH;
    if (hipFuncSetAttribute((const void*)h2_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDwLds) != hipSuccess) return LGD_ELAUNCH;
    if (hipFuncSetAttribute((const void*)h2_pwdw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPwDwLds) != hipSuccess) return LGD_ELAUNCH;
    done[dev] = true;
    return LGD_OK;
}

// CUs of the current device, asked once per device (hipGetDeviceProperties is slow on ROCm and the split helpers sit on the host's launch path:
// once per 3x3 and 1x1 weight-gradient product, ~80 calls per step -- ADVICE r5)
int cu_count() {
    static int cus_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus_dev[dev]) {
        hipDeviceProp_t prop;
        cus_dev[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus_dev[dev];
}

}  // namespace
}  // namespace lgd

extern "C" {

size_t lgd_h2_image_bytes(int nb, int M, int K) {
    if (nb <= 0 || M <= 0 || K <= 0) return 0;
    return (size_t)nb * ((K + 15) / 16) * 2 * ((M + 31) / 32) * 1024;
}

int lgd_h2_fwd(const void* image, const void* B, long long b_sb, long long b_sk, long long b_bytes, float* C, long long c_sb, long long c_sm,
               const float* a_inv, const float* b_inv, int b_inv_per_batch, uint32_t* amax_out, int nb, int M, int N, int K, void* stream) {
    if (!image || !B || !C || !a_inv || !b_inv || nb <= 0 || M <= 0 || N <= 0 || K <= 0 || (K & 15) || ((uintptr_t)image & 15) || ((uintptr_t)B & 15) ||
        (b_sb & 15) || (b_sk & 15) || b_sk <= 0 || b_sb < 0 || c_sm <= 0)
        return LGD_EINVAL;
    // 32-bit byte offsets from the descriptors' origins inside the kernel: 16 rows of B, one tile of C rows
    if (16 * b_sk + 4LL * N >= (1LL << 31) || 256LL * c_sm >= (1LL << 29)) return LGD_EINVAL;
    if (lgd::ensure_attrs() != LGD_OK) return LGD_ELAUNCH;
    lgd::FwdP p;
    p.rbp = (M + 31) / 32;
    p.Aimg = (const char*)image; p.a_sb = (long)(K / 16) * 2 * p.rbp * 1024;
    p.B = (const char*)B; p.b_sb = (long)b_sb; p.b_ld = (long)b_sk; p.b_bytes = (long)b_bytes;
    p.C = C; p.c_sb = (long)c_sb; p.c_ld = (long)c_sm; p.a_inv = a_inv; p.b_inv = b_inv; p.b_inv_stride = b_inv_per_batch ? 1 : 0; p.amax_out = amax_out;
    const bool small = ((M + 255) / 256 * 256 - M >= 64 && (M + 127) / 128 * 128 - M < 64);
    const int bm = small ? 128 : 256;
    p.nb = nb; p.M = M; p.N = N; p.K = K; p.mt = (M + bm - 1) / bm; p.nt = (N + 127) / 128;
    const unsigned total = (unsigned)(((nb + 7) / 8) * p.nt * p.mt * 8);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(total), block(256);
    if (small) {
        if (amax_out) { LGD_LAUNCH("h2_fwd_kernel", (lgd::h2_fwd_kernel<128, true>), grid, block, lgd::kFwdLds128, st, p); }
        else { LGD_LAUNCH("h2_fwd_kernel", (lgd::h2_fwd_kernel<128, false>), grid, block, lgd::kFwdLds128, st, p); }
    } else {
        if (amax_out) { LGD_LAUNCH("h2_fwd_kernel", (lgd::h2_fwd_kernel<256, true>), grid, block, lgd::kFwdLds256, st, p); }
        else { LGD_LAUNCH("h2_fwd_kernel", (lgd::h2_fwd_kernel<256, false>), grid, block, lgd::kFwdLds256, st, p); }
    }
    return lgd::check_launch();
}

int lgd_h2_dw_splits(int nb, int M, int N, int T) {
    if (nb <= 0 || M <= 0 || N <= 0 || T < 16) return 1;
    const int cus = lgd::cu_count();
    const long tiles = (long)nb * ((M + 255) / 256) * ((N + 255) / 256);
    const int nstage = T / 16;
    // one workgroup per CU: as many splits as fill the chip once (measured: 64 batches x 4 on 256 CUs 184 us, x 2 246, x 8 224), at least 8 stages each
    int S = (int)((cus + tiles - 1) / tiles);
    if (S < 1) S = 1;
    while (S > 1 && nstage / S < 8) --S;
    return S > 64 ? 64 : S;
}

int lgd_h2_dw(const void* A, long long a_rs, long long a_sb, long long a_bytes, const float* a_inv, int a_inv_per_batch, const void* B, long long b_rs,
              long long b_sb, long long b_bytes, const float* b_inv, int b_inv_per_batch, float* out, float* partials, int S, int nb, int M, int N, int T,
              void* stream) {
    if (!A || !B || (!out && S == 1) || !a_inv || !b_inv || nb <= 0 || M <= 0 || N <= 0 || T < 16 || (T & 31) || S < 1 || (S > 1 && !partials) ||
        ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || (a_rs & 15) || (b_rs & 15) || (a_sb & 15) || (b_sb & 15) || a_rs <= 0 || b_rs <= 0)
        return LGD_EINVAL;
    if (32 * a_rs >= (1LL << 31) || 32 * b_rs >= (1LL << 31)) return LGD_EINVAL;   // 32-bit byte offsets across a wave's 32 rows
    if (lgd::ensure_attrs() != LGD_OK) return LGD_ELAUNCH;
    lgd::DwP p;
    p.A = (const char*)A; p.a_rs = (long)a_rs; p.a_sb = (long)a_sb; p.a_bytes = (long)a_bytes;
    p.B = (const char*)B; p.b_rs = (long)b_rs; p.b_sb = (long)b_sb; p.b_bytes = (long)b_bytes;
    p.P = S > 1 ? partials : out; p.a_inv = a_inv; p.b_inv = b_inv; p.a_inv_stride = a_inv_per_batch ? 1 : 0; p.b_inv_stride = b_inv_per_batch ? 1 : 0;
    p.nb = nb; p.M = M; p.N = N; p.nstage = T / 16; p.S = S; p.per = (p.nstage + S - 1) / S;
    p.mt = (M + 255) / 256; p.nt = (N + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    LGD_LAUNCH("h2_dw_kernel", lgd::h2_dw_kernel, dim3((unsigned)(nb * S * p.mt * p.nt)), dim3(512), lgd::kDwLds, st, p);
    if (S > 1 && out) {   // (out == NULL: the partials are the result -- lgd_wino_filter_bwd_parts adds them while it reads)
        const long n = (long)nb * M * N;
        if (n & 3) return LGD_EINVAL;
        LGD_LAUNCH("h2_reduce_kernel", lgd::h2_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, partials, out, n / 4, S);
    }
    return lgd::check_launch();
}

int lgd_h2_pwdw_splits(int nimg, int M, int N, int HW) {
    if (nimg <= 0 || M <= 0 || N <= 0 || HW <= 0) return 1;
    const int cus = lgd::cu_count();
    const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256), nstage = (long)nimg * ((HW + 31) / 32);
    long S = (cus + tiles - 1) / tiles;           // one workgroup per CU
    while (S > 1 && nstage / S < 8) --S;
    return (int)(S < 1 ? 1 : S);
}

// partials[s] (M x N, row-major) = sum over split s of the (image, pixel) range of dz[n][m][px] x[n][k][px]; the caller adds the S partials
// (lgd_sum_batch_scale: fixed order, with the frozen per-row scale).  a_amax / b_amax: bounds of |dz| / |x| (float bits).
int lgd_h2_pwdw(const float* dz, const float* x, const uint32_t* a_amax, const uint32_t* b_amax, float* partials, int S, int nimg, int M, int N, int HW,
                void* stream) {
    if (!dz || !x || !a_amax || !b_amax || !partials || S < 1 || nimg < 1 || M < 1 || N < 1 || HW < 4 || (HW & 3) || ((uintptr_t)dz & 15) || ((uintptr_t)x & 15))
        return LGD_EINVAL;
    if (lgd::ensure_attrs() != LGD_OK) return LGD_ELAUNCH;
    lgd::PwDwP p;
    p.A = dz; p.B = x; p.P = partials; p.a_amax = a_amax; p.b_amax = b_amax;
    p.nimg = nimg; p.M = M; p.N = N; p.HW = HW; p.spi = (HW + 31) / 32; p.nstage = nimg * p.spi; p.S = S; p.per = (p.nstage + S - 1) / S;
    p.mt = (M + 255) / 256; p.nt = (N + 255) / 256;
    LGD_LAUNCH("h2_pwdw_kernel", lgd::h2_pwdw_kernel, dim3((unsigned)(S * p.mt * p.nt)), dim3(512), lgd::kPwDwLds, (hipStream_t)stream, p);
    return lgd::check_launch();
}

int lgd_h2_amax_maps(const float* const* x_host, const int32_t* level_hw_host, int L, int N, int C, const float* pre_bias, const float* pre_affine,
                     uint32_t* out_bits, int accumulate, void* stream) {
    if (!x_host || !level_hw_host || !out_bits || L < 1 || L > LGD_MAX_LEVELS || N < 1 || C < 1 || (pre_bias && pre_affine)) return LGD_EINVAL;
    lgd::AmaxArgs a{};
    unsigned blk = 0;
    for (int l = 0; l < L; ++l) {
        const long hw = (long)level_hw_host[2 * l] * level_hw_host[2 * l + 1];
        if (!x_host[l] || hw < 1 || hw >= (1L << 31)) return LGD_EINVAL;
        a.maps[l] = x_host[l]; a.hw[l] = (int)hw; a.cpp[l] = (int)((hw + lgd::kAmaxChunk - 1) / lgd::kAmaxChunk);
        a.chunk[l] = (int)(((hw + a.cpp[l] - 1) / a.cpp[l] + 3) & ~3L);
        a.blk_off[l] = blk;
        const long nblk = (long)N * C * a.cpp[l];
        if (blk + nblk >= (1L << 31)) return LGD_EINVAL;
        blk += (unsigned)nblk;
    }
    a.blk_off[L] = blk;
    a.bias = pre_bias; a.affine = pre_affine; a.out = out_bits; a.L = L; a.N = N; a.C = C;
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate && hipMemsetAsync(out_bits, 0, sizeof(uint32_t), st) != hipSuccess) return LGD_ELAUNCH;
    LGD_LAUNCH("h2_amax_maps_kernel", lgd::h2_amax_maps_kernel, dim3(blk), dim3(256), 0, st, a);
    return lgd::check_launch();
}

int lgd_h2_amax_filters(const float* const* w_host, const float* const* scale_host, const int32_t* rows_host, int K, int row_elems, uint32_t* out_bits,
                        void* stream) {
    if (!w_host || !rows_host || !out_bits || K < 1 || K > 8 || row_elems < 1) return LGD_EINVAL;
    lgd::AmaxFilterArgs a{};
    unsigned off = 0;
    for (int k = 0; k < K; ++k) {
        if (!w_host[k] || rows_host[k] < 1) return LGD_EINVAL;
        a.w[k] = w_host[k]; a.scale[k] = scale_host ? scale_host[k] : nullptr; a.rows[k] = rows_host[k]; a.row_off[k] = off;
        off += (unsigned)rows_host[k];
    }
    a.row_off[K] = off; a.out = out_bits; a.K = K; a.row = row_elems;
    hipStream_t st = (hipStream_t)stream;
    LGD_LAUNCH("h2_amax_filter_kernel", lgd::h2_amax_filter_kernel, dim3(off), dim3(256), 0, st, a);
    return lgd::check_launch();
}

int lgd_h2_words_max(const uint32_t* const* words_host, int n, uint32_t* out, void* stream) {
    if (!words_host || !out || n < 1 || n > 16) return LGD_EINVAL;
    lgd::WordsArgs a{};
    for (int i = 0; i < n; ++i) { if (!words_host[i]) return LGD_EINVAL; a.w[i] = words_host[i]; }
    a.n = n; a.out = out;
    LGD_LAUNCH("h2_words_max_kernel", lgd::h2_words_max_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

int lgd_h2_link_bound(const uint32_t* amax64, uint32_t* out_bits, void* stream) {
    if (!amax64 || !out_bits) return LGD_EINVAL;
    LGD_LAUNCH("h2_link_bound_kernel", lgd::h2_link_bound_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, amax64, out_bits);
    return lgd::check_launch();
}

int lgd_h2_plane_sums(const void* buf, long long ch_bytes, int C, int T, const float* inv, float* out, void* stream) {
    if (!buf || !inv || !out || C < 1 || T < 32 || T % 32 || ch_bytes < 4LL * T) return LGD_EINVAL;
    LGD_LAUNCH("h2_plane_sums_kernel", lgd::h2_plane_sums_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, (const char*)buf, ch_bytes, T, inv, out);
    return lgd::check_launch();
}

}  // extern "C"
