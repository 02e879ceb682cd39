// K9: the channel products of the Winograd convolutions on the bf16 MFMA pipe, fp32 in HBM on both sides  (include/lgd_hip.h: lgd_gemm3*)
//   C[b] (M x N) = A[b] (M x K) . B[b] (K x N)      [the arithmetic of nn.Conv2d(C, C', 3, padding=1): dynamic_teacher.py:57,61,67-73,
//                                                     145,280; sequential_convs.py:10-12; the head towers, distillator.py:107-109]
// gfx950 runs fp32-input MFMA at the VECTOR rate (157 TFLOP/s, 1/16 of bf16; the xf32 forms of gfx942 are gone) and the library's
// fp32 GEMMs were 68 % of the training step.  Here every fp32 element x is split into three bf16 pieces, x = h + m + l with
// h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest: the sum is x to 2^-24 |x|), and the product keeps 6 of the 9
// cross terms:  a b ~ ah bh + (ah bm + am bh) + (ah bl + am bm + al bh);  the dropped ones are <= ~2^-23 |a||b|, accumulation is fp32
// inside v_mfma_f32_32x32x16_bf16.  Measured against an fp64 product of the same operands: 6.2e-7 of the output scale (rocBLAS fp32:
// 7.6e-7).  A two-piece split (2^-16) does not survive the F(6x6,3x3) transforms; three pieces do.
//
// What keeps it an MFMA-bound kernel instead of a VALU-bound one (lab history: tools/lab/gemm3_lab.hip, profiles/r04_gemm3_lab*.log):
//  * the FILTER operand A is split once, ahead of the product (split_a_kernel), into an image in MFMA fragment order --
//    [batch][k-step of 16][piece][32-row block][lane][8 bf16] -- so that a k-step's share of a 256-row tile is 3 runs of 8 KB which
//    LDS-DMA (global_load_lds_dwordx4: no registers, no VALU) drops into LDS unchanged;
//  * the workgroup's tile spans 256 rows of A (all output channels of a 256 -> 256 convolution), so every element of B (the
//    activations, 342 MB per product at BASELINE config 2) is read from HBM once and split once: ~1.4 VALU instructions per MFMA
//    (splitting both operands in a 128 x 128 tile: 7.8, issue-bound at 30 % of the pipe);
//  * all addresses of the k-loop are a uniform base + per-thread 32-bit offsets computed once.
// Tile 256 x 128 x 16, 256 threads (2 x 2 waves, 128 x 64 per wave = 4 x 2 MFMA blocks of 32 x 32, 128 accumulator registers), two
// LDS buffers of 36 KB (A pieces 24 KB + B pieces 12 KB): one barrier per k-step, two workgroups per CU so that one's staging and
// epilogue hide under the other's MFMAs.  A 128-row variant of the same kernel serves C' = 128 layers.  The image may be shared by all
// batches (the student's 1x1 convolutions: one filter, a batch of images).  The 128-row kernel also carries the bottleneck blocks'
// epilogue: C = relu?(A B + R + shift[m]) with the accumulators INITIALISED from the residual map R and the per-row shift (R = C: the
// accumulating form C += A B), the ReLU applied on the way out and its 1-bit mask written from wave ballots (one word per 32 columns of
// a row) -- the pre-activation map is neither written nor re-read by a bias / residual / ReLU pass.
#include <stdlib.h>

#include <type_traits>

#include "winograd.h"   // h2_exponent / h2_pow2: the power-of-two scales of the f16x2 form

namespace lgd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef lgd_u32x4 u32x4;

constexpr int BN = 128, BK = 16, NT = 256;
#ifndef LGD_GEMM3_PIPE
#define LGD_GEMM3_PIPE 1   // 1: staging behind the k-step's barrier (shipped); 0: at the top of the k-step (lab: the form until the end of round 4)
#endif
#ifndef LGD_GEMM3_ABL
#define LGD_GEMM3_ABL 0   // lab ablations of the staging parts (tools/gpu_checks.sh ablate; results are garbage): 1 no split arithmetic, 2 no B at
#endif                    // all, 3 no image DMA, 4 no C stores, 5 none of them (MFMA phase, fragment reads and barriers only)

// tile rows BM = 256 (4 x 2 MFMA blocks per wave, 2 workgroups per CU) or 128 (2 x 2 blocks, 64 accumulator registers, 3 workgroups per
// CU: C' = 128 layers, whose 256-row tile would idle half of every MFMA)
// PCS: pieces per operand -- 3: bf16 (x = h + m + l, six of nine cross products); 2: f16 (x 2^e = h + m, three of four: the f16x2 form of
// csrc/h2.hip with the activation operand split in registers -- the student's 1x1 convolutions, whose maps arrive as fp32; round 5)
template <int BM, int PCS = 3> struct Tile {
    static constexpr int RB = BM / 32, MI = BM / 64;                      // 32-row blocks per tile / per wave
    static constexpr int A_BYTES = PCS * RB * 1024, B_BYTES = PCS * 4 * 1024, BUF = A_BYTES + B_BYTES, LDS_BYTES = 2 * BUF;   // 72 / 48 KB (48 / 32)
    static constexpr int CHUNKS = PCS * RB / 4;                           // 1 KB LDS-DMA pieces per wave and k-step
};

#define LGD_GLDS16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(const f16x8& a, const f16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// two (already scaled) elements -> packed h pair, packed m pair of the f16x2 form (csrc/h2.hip)
__device__ __forceinline__ void split2_f16(float t0, float t1, uint32_t& h, uint32_t& m) {
    typedef _Float16 hx2 __attribute__((ext_vector_type(2)));
    const hx2 hh = __builtin_convertvector((lgd_f32x2){t0, t1}, hx2);
    const hx2 mm = __builtin_convertvector((lgd_f32x2){t0 - (float)hh[0], t1 - (float)hh[1]}, hx2);
    h = __builtin_bit_cast(uint32_t, hh);
    m = __builtin_bit_cast(uint32_t, mm);
}

struct Params {
    const char* Aimg; long a_sb; int rbp, ktp;   // image [nb][ktp][3][rbp][1024 B]; a_sb in bytes
    const float* B; long b_sb, b_ld;             // B(k, n) at B[k * b_ld + n]
    float* C; long c_sb, c_ld;                   // C(m, n) at C[m * c_ld + n]
    int nb, M, N, K, mt, nt;
    int total;                                   // workgroup ids to walk (tiles, rounded up to a multiple of 8)
    // EPI kernels: C = relu?(A B + R + shift[m]); R (may be C itself), shift and bits may each be NULL
    const float* R; long r_sb, r_ld, r_bytes;    // r_bytes: extent of R from its first element (the buffer descriptor's range)
    const float* shift;
    uint32_t* bits; int wpr;                     // ReLU mask: bit n % 32 of word (b * M + m) * wpr + n / 32 = (C(m, n) > 0)
    int relu;
    const float* a_inv; const unsigned* b_amax;  // PCS == 2: inverse scale of the image (one per image), bound of |B| (float bits): B is scaled by 2^eb in the kernel
    int kper; long part_stride;                  // split-K (gridDim.y > 1): k-steps per split; split s writes its partial product to C + s * part_stride
    unsigned* amax;                              // optional: atomic max of the float bits of |C| as stored (one word, zeroed by the caller): the bound
                                                 // the NEXT convolution's f16x2 scale is derived from (csrc/h2.hip)
};

// EPI: 0 the plain product; otherwise the epilogue kernel with bit 0: R present, bit 1: shift present
// waves per SIMD the register allocator is held to: three workgroups per CU for the 128-row tile (left alone the epilogue forms take 172-180
// registers -- two workgroups; tools/gemm2h_probe.py: res3's 512 -> 128 convolution with its epilogue 124 -> 101 us).  The f16x2 form only:
// two of the bf16x3 instances would spill at 168, and nothing large runs on them any more.
template <int BM, int EPI, int PCS> constexpr int gemm3_waves() { return BM == 128 && PCS == 2 ? 3 : 2; }
template <int BM, int EPI, int PCS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(gemm3_waves<BM, EPI, PCS>()))) void gemm3_kernel(const Params p) {
    typedef Tile<BM, PCS> TL;
    typedef typename std::conditional<PCS == 3, bf16x8, f16x8>::type frag_t;
    constexpr int A_BYTES = TL::A_BYTES, BUF = TL::BUF, MI = TL::MI, RB = TL::RB, CH = TL::CHUNKS;
    extern __shared__ __attribute__((aligned(1024))) char lds[];   // ONE LDS object (a second one makes hipcc drain vmcnt(0) before every ds_read)
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 1, wn = w & 1;
    // workgroup -> (batch, n-tile, m-tile).  Consecutive ids go round-robin to the 8 XCDs: XCD x takes the batches b = x (mod 8) and walks
    // their tiles in order, so that its L2 holds the images of the one or two batches it is working on (tiles of one batch spread over
    // all XCDs: every L2 holds all ~13 images in flight; measured 2 % slower)
    // A workgroup walks the tiles id, id + gridDim, ... (gridDim a multiple of 8: all on one XCD): launched with one workgroup per tile
    // it runs the body once; launched PERSISTENT (as many workgroups as fit the chip at once) the next tile's prologue is issued while
    // the last tile's stores drain, and no slot waits for a workgroup to retire and another to be dispatched.
    for (int id = blockIdx.x; id < p.total; id += gridDim.x) {
    // (persistent launch only) the previous tile's stores may be acknowledged out of order with the loads below: the k-loop's counted
    // waits need an empty memory pipe in front of the first load of a tile
    if (id != (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int xcd = id & 7, j = id >> 3;
    int b, tn, sub;
    if (p.a_sb != 0) {
        const int per_b = p.nt * p.mt;
        b = (j / per_b) * 8 + xcd;
        const int r = j % per_b;
        if (b >= p.nb) continue;
        tn = r / p.mt; sub = r % p.mt;
    } else {   // ONE image for all batches: nothing ties a batch to an XCD; the m-tiles that read the same B tile stay neighbours on one L2
        sub = j % p.mt;
        const int rest = (j / p.mt) * 8 + xcd;
        if (rest >= p.nb * p.nt) continue;
        b = rest / p.nt; tn = rest % p.nt;
    }
    const int m0 = sub * BM, n0 = tn * BN, rb0 = sub * RB;
    // split-K: workgroup row blockIdx.y takes the k-steps [ks0, ks0 + ksteps) and leaves a PARTIAL product (plain launches only: host-checked)
    // (the epilogue kernels are never split: the code is compiled out of them)
    const int ks0 = EPI == 0 ? (int)blockIdx.y * p.kper : 0;
    const int ksteps = EPI == 0 ? min(p.kper, p.K / BK - ks0) : p.K / BK;    // K % 16 == 0 (host-checked)
    const char* Ai = p.Aimg + (long)b * p.a_sb + (long)ks0 * ((long)PCS * p.rbp * 1024);
    // B staging: thread <-> (k-group kg = t >> 7, column n = t & 127): 8 dwords down the k axis, a wave's load covers 256 contiguous
    // bytes.  Columns >= N are read from column N - 1 and rows >= M come as zeros from the image: both only reach elements of C that are
    // never stored (rows and columns of a product are independent), so nothing is zeroed here.
    const int kg = t >> 7, nl = t & 127;
    const int ncol = n0 + nl < p.N ? n0 + nl : p.N - 1;
    const float* Bb = p.B + (long)b * p.b_sb + (long)ks0 * BK * p.b_ld;
    uint32_t boff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) boff[e] = (uint32_t)((kg * 8 + e) * (int)p.b_ld + ncol) * 4u;   // bytes
    const long bstep = (long)BK * p.b_ld;
    const int bslot = A_BYTES + (nl >> 5) * 1024 + kg * 512 + (nl & 31) * 16;   // 8 consecutive lanes -> 128 contiguous bytes: no conflicts
    // PCS == 2: power-of-two scale of B from its bound; the total exponent stays <= 100 so that an accumulator started from R 2^(ea + eb) cannot
    // overflow when the filter is (nearly) zero: a zero-initialised layer has ea = 126
    float bscale = 1.f, inv_tot = 1.f, sc_tot = 1.f;
    if constexpr (PCS == 2) {
        const float ainv = p.a_inv[p.a_sb != 0 ? b : 0];
        const int ea = 127 - (int)((__builtin_bit_cast(unsigned, ainv) >> 23) & 0xffu);
        int eb = h2_exponent(*p.b_amax, 0);
        eb = eb > 100 - ea ? 100 - ea : eb;
        bscale = h2_pow2(eb);
        sc_tot = h2_pow2(ea + eb);
        inv_tot = h2_pow2(-(ea + eb));
    }
    float bv[8];
    // buffer loads: the k-step's origin in the descriptor (scalar), the eight per-thread offsets as they are -- no 64-bit address per load
    auto load_b = [&](int ks) {
#if LGD_GEMM3_ABL == 2 || LGD_GEMM3_ABL == 5
        return;
#endif
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bb + ks * bstep), 0, 0xffffffffu, 0x00020000);
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)boff[e], 0, 0));
    };
    auto store_b = [&](char* buf) {
#if LGD_GEMM3_ABL == 2 || LGD_GEMM3_ABL == 5
        return;
#endif
        uint32_t h[4], m[4], l[4];
        char* d = buf + bslot;
        if constexpr (PCS == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) split2_f16(bv[2 * e] * bscale, bv[2 * e + 1] * bscale, h[e], m[e]);
            *reinterpret_cast<u32x4*>(d) = (u32x4){h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4*>(d + 4096) = (u32x4){m[0], m[1], m[2], m[3]};
            return;
        }
#if LGD_GEMM3_ABL == 1   // lab: the LDS store without the split's arithmetic
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = m[e] = l[e] = __builtin_bit_cast(uint32_t, bv[2 * e]) ^ __builtin_bit_cast(uint32_t, bv[2 * e + 1]);
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) split2(bv[2 * e], bv[2 * e + 1], h[e], m[e], l[e]);
#endif
        *reinterpret_cast<u32x4*>(d) = (u32x4){h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(d + 4096) = (u32x4){m[0], m[1], m[2], m[3]};
        *reinterpret_cast<u32x4*>(d + 8192) = (u32x4){l[0], l[1], l[2], l[3]};
    };
    // A: 3 * RB chunks of 1 KB per k-step ([piece][row block]); wave w moves chunks CH w .. CH w + CH - 1 (row blocks past the image repeat
    // its last one: their products are rows >= M)
    uint32_t aoff[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int ch = w * CH + c, pc = ch / RB, rbl = ch % RB;
        int rb = rb0 + rbl;
        rb = rb < p.rbp ? rb : p.rbp - 1;
        aoff[c] = (uint32_t)((pc * p.rbp + rb) * 1024 + lane * 16);
    }
    const long astep = (long)PCS * p.rbp * 1024;
    auto dma_a = [&](int ks, char* buf) {
#if LGD_GEMM3_ABL == 3 || LGD_GEMM3_ABL == 5
        return;
#endif
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ai + ks * astep), 0, 0xffffffffu, 0x00020000);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(buf + (w * CH + c) * 1024), 16, (int)aoff[c], 0, 0, 0);
    };
    // the accumulators start from zero, or (EPI) from R + shift: the MFMA chain adds the product on top and the epilogue is the plain
    // store (an epilogue that re-reads a map needs the 128 accumulators in VGPRs at once: one resident workgroup instead of two)
    const int g = lane >> 5, rr = lane & 31;
    const int mw = m0 + wm * (BM / 2) + 4 * g, nw = n0 + wn * 64 + rr;
    const int ld = (int)p.c_ld;
    const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
    f32x16 acc[MI][2];
    if constexpr (EPI != 0) {
        // R by BUFFER loads straight into the accumulator registers, all of a lane's loads in flight at once: one per-lane 32-bit offset
        // for the whole tile + a scalar offset per row (global loads took a 64-bit address pair per row: 32 VGPRs, the third resident
        // workgroup), and the range check of the descriptor stands in for every branch -- rows past M and columns past N read other
        // elements of the map or, past its end, zeros: values that only reach elements never stored.  The shift is added on the way out.
        const int wrow = m0 + wm * (BM / 2), wcol = n0 + wn * 64;
        const int rld = (int)p.r_ld;
        __amdgpu_buffer_rsrc_t rr_src;
        if constexpr ((EPI & 1) != 0) {
            const long o = ((long)b * p.r_sb + (long)wrow * rld + wcol) * 4;
            const long left = p.r_bytes - o;
            rr_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>((const char*)p.R + o), 0, (uint32_t)(left < 0 ? 0 : left > 0xffffffffL ? 0xffffffffL : left), 0x00020000);
        }
        // the row offset runs in ONE register (+ 1 or + 5 rows per step): as scalar offsets the compiler precomputes all 128 at kernel entry,
        // spills them into VGPR lanes and pays v_readlane + s_nop 4 in front of every load
        int ro = (4 * g * rld + rr) * 4;
        const int r1 = rld * 4, r5 = rld * 20;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if constexpr ((EPI & 1) != 0) {
                    acc[i][0][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr_src, ro, 0, 0));
                    acc[i][1][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr_src, ro + 128, 0, 0));
                    if constexpr (PCS == 2) { acc[i][0][e] *= sc_tot; acc[i][1][e] *= sc_tot; }   // (the product on top of it is scaled by 2^(ea + eb))
                    ro += (e & 3) == 3 ? r5 : r1;
                    asm volatile("" : "+v"(ro));
                } else {
                    acc[i][0][e] = 0.f;
                    acc[i][1][e] = 0.f;
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;
    }

    // the accumulators live in the accumulator file from here on (zero-initialised ones left to the compiler: it folds the zeros into the
    // first MFMAs, peels the k-loop's first step and, in the paired loop, holds 100 more VGPRs)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) asm volatile("" : "+a"(acc[i][jn]));
    const int slot = lane * 16;
#if LGD_GEMM3_PIPE == 1
    // Staging shifted by half a k-step.  hipcc hoists the k-step's barrier (with its s_waitcnt vmcnt(0): the LDS-DMA must have landed) to
    // the point where the last fragment reads are issued -- in front of the last 24 MFMAs.  Staged at the TOP of the k-step, the image
    // DMA and the B loads had half a k-step (~0.35 us) to come back from L2 / HBM before that wait; staged right BEHIND the barrier --
    // into the buffer whose fragments every wave has just taken, for the k-step after next -- they have a whole one: -2.7 % on the
    // two-pyramid products (508 -> 494 us, profiles/r04_gemm3_pipe.log).  B a SECOND k-step ahead (two register sets, the loop in pairs)
    // measured the same as this: memory latency is not what the waves wait for any more.
    dma_a(0, lds);
    load_b(0);
    store_b(lds);
    if (ksteps > 1) {
        load_b(1);
        dma_a(1, lds + BUF);
        store_b(lds + BUF);
    }
    load_b(ksteps > 2 ? 2 : ksteps - 1);   // always 8 loads behind the youngest DMA: the counted wait below
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        char* cur = lds + (ks & 1) * BUF;
        frag_t fb[PCS][2], fa[MI];
#pragma unroll
        for (int pc = 0; pc < PCS; ++pc)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
                fb[pc][jn] = *reinterpret_cast<const frag_t*>(cur + A_BYTES + pc * 4096 + (wn * 2 + jn) * 1024 + slot);
#pragma unroll
        for (int pa = PCS - 1; pa >= 1; --pa) {      // smallest pieces first
#pragma unroll
            for (int i = 0; i < MI; ++i)
                fa[i] = *reinterpret_cast<const frag_t*>(cur + pa * (RB * 1024) + (wm * MI + i) * 1024 + slot);
#pragma unroll
            for (int pb = PCS - 1 - pa; pb >= 0; --pb)   // pa + pb <= PCS - 1: the six (three) kept products
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        acc[i][jn] = mfma16(fa[i], fb[pb][jn], acc[i][jn]);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
            fa[i] = *reinterpret_cast<const frag_t*>(cur + (wm * MI + i) * 1024 + slot);
        // the image DMA of k-step ks + 1 (issued behind the previous barrier) must have LANDED before anybody passes this one; hipcc's own
        // wait insertion loses an LDS-DMA across the loop's back-edge (it emitted no vmcnt wait here at all), and vmcnt(0) would also
        // drain the 8 B loads issued behind that DMA, which nobody needs before store_b.  The memory pipe returns in order: vmcnt(8).
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();   // every wave holds its last fragments of cur; k-step ks + 1 is complete in the other buffer
        if (ks + 2 < ksteps) {
            store_b(cur);                            // bv: B of k-step ks + 2, loaded behind the previous barrier
            dma_a(ks + 2, cur);
        }
        if (ks + 1 < ksteps) load_b(ks + 3 < ksteps ? ks + 3 : ksteps - 1);   // (the last ones re-read a valid k-step: the count stays 8)
#pragma unroll
        for (int pb = PCS - 1; pb >= 0; --pb)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
                    acc[i][jn] = mfma16(fa[i], fb[pb][jn], acc[i][jn]);
    }
    __syncthreads();       // (the epilogue and the next tile's prologue reuse the buffers)
#else
    static_assert(PCS == 3, "the lab form of the k-loop exists for bf16x3 only");
    // prologue: k-step 0 into buffer 0, B of k-step 1 into registers
    dma_a(0, lds);
    load_b(0);
    store_b(lds);
    if (ksteps > 1) load_b(1);
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        char* cur = lds + (ks & 1) * BUF;
        char* nxt = lds + ((ks + 1) & 1) * BUF;
        if (ks + 1 < ksteps) {
            // order matters to hipcc's wait insertion: the use of bv (loaded a whole k-step ago) comes BEFORE the LDS-DMA is issued --
            // with a DMA in flight the compiler waits vmcnt(0) at the next use of an ordinary load's result, which would expose the DMA.
            // (A hand-counted variant -- asm loads two k-steps ahead, counted vmcnt(14) / vmcnt(8) -- measured 281 against 285 us and
            //  produced wrong tiles under load: hipcc may copy an asm load's destination at the loop back-edge before the data lands.)
            store_b(nxt);
            dma_a(ks + 1, nxt);
            if (ks + 2 < ksteps) load_b(ks + 2);
        }
        bf16x8 fb[3][2];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
                fb[pc][jn] = *reinterpret_cast<const bf16x8*>(cur + A_BYTES + pc * 4096 + (wn * 2 + jn) * 1024 + slot);
#pragma unroll
        for (int pa = 2; pa >= 0; --pa) {            // smallest pieces first
            bf16x8 fa[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(cur + pa * (RB * 1024) + (wm * MI + i) * 1024 + slot);
#pragma unroll
            for (int pb = 2 - pa; pb >= 0; --pb)     // pa + pb <= 2: the six kept products
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[pb][jn], acc[i][jn], 0, 0, 0);
        }
        __syncthreads();
    }
#endif
    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); 32-bit offsets from one base;
    // a half-wave's store covers 128 contiguous bytes.  EPI: shift and ReLU on the way out, straight-line per (mask?, full tile?) variant
    // (a condition per element splits the block into 16 basic blocks with a wait each).  The ReLU mask: a ballot per accumulator register
    // is two words (the low one row m's 32 columns, the high one row m + 4's); the 32 words of a 32 x 32 block are dropped into lanes
    // 0..31 of one register (v_writelane) and leave as ONE store per block -- lane j holds (e = j & 15, half = j >> 4).
    {
        const bool norelu = p.relu == 0;
        const int wrow = m0 + wm * (BM / 2), wcol = n0 + wn * 64;
        // the tile's shift values through LDS (free after the k-loop's last barrier): a global load per row here would queue behind the
        // block's own stores (gfx9 counts loads and stores in one in-order vmcnt) and wait for their acknowledgements, block after block
        const float* lsh = reinterpret_cast<const float*>(lds) + wm * (BM / 2) + 4 * g;
        if constexpr ((EPI & 2) != 0) {
            if (t < BM) reinterpret_cast<float*>(lds)[t] = m0 + t < p.M ? p.shift[m0 + t] : 0.f;
            __syncthreads();
        }
        const int brow = ((lane & 3) + 8 * ((lane & 15) >> 2) + 4 * ((lane >> 4) & 1));   // the row (within a block) of the word lane j < 32 holds
        uint32_t* bwl = p.bits + ((long)b * p.M + wrow + brow) * p.wpr + (wcol >> 5);
        // C by buffer stores: the wave's origin in the descriptor, ONE running per-lane offset (+ 1 or + 5 rows per step; as scalar offsets
        // hipcc precomputes all of them at kernel entry, spills them into VGPR lanes and pays a v_readlane per store)
        const __amdgpu_buffer_rsrc_t cs = __builtin_amdgcn_make_buffer_rsrc(p.C + (EPI == 0 ? (long)blockIdx.y * p.part_stride : 0L) + (long)b * p.c_sb + (long)wrow * ld + wcol, 0, 0xffffffffu, 0x00020000);
        const int c1 = ld * 4, c5 = ld * 20, mrem = p.M - mw;
        const bool colok[2] = {nw < p.N, nw + 32 < p.N};
        const bool want_max = p.amax != nullptr;
        uint32_t amax = 0u;
        auto epi = [&](auto HB, auto HF) {
            constexpr bool hb = decltype(HB)::value, hf = decltype(HF)::value;
            int cbase = (4 * g * ld + rr) * 4;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    asm volatile("" : "+a"(acc[i][jn]) :: "memory");   // one block out of the accumulator file at a time (all at once: 64+ VGPRs)
                    int words = 0, co = cbase;
                    uint32_t bm = 0u;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int dm = i * 32 + (e & 3) + 8 * (e >> 2);
                        float v = acc[i][jn][e];
                        if constexpr (PCS == 2) v *= inv_tot;
                        if constexpr ((EPI & 2) != 0) v += lsh[dm];
                        if constexpr (EPI != 0) {
                            const bool pos = v > 0.f;
                            if constexpr (hb) {
                                const unsigned long long bal = __ballot(pos);
                                // s_nop: v_writelane reads a stale SGPR when it issues right behind the VALU compare that wrote it (measured: every
                                // word held the previous register's ballot); the hazard recogniser does not look inside inline asm
                                asm("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(words) : "s"((uint32_t)bal), "n"(e));
                                asm("v_writelane_b32 %0, %1, %2" : "+v"(words) : "s"((uint32_t)(bal >> 32)), "n"(16 + e));
                            }
                            v = (pos | norelu) ? v : 0.f;
                        }
#if LGD_GEMM3_ABL == 4 || LGD_GEMM3_ABL == 5   // lab: no C stores (one conditional store keeps the accumulators alive)
                        if (v == 123456.f)
#endif
                        const bool ok = hf || (dm < mrem && colok[jn]);
                        if (ok) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), cs, co + jn * 128, 0, 0);
                        bm = max(bm, ok ? __builtin_bit_cast(uint32_t, v) & 0x7fffffffu : 0u);   // (always: a branch per element costs registers, not time)
                        co += (e & 3) == 3 ? c5 : c1;
                        asm volatile("" : "+v"(co));
                    }
                    if constexpr (hb) {
                        if (lane < 32 && wrow + i * 32 + brow < p.M && wcol + jn * 32 < p.N) bwl[(long)(i * 32) * p.wpr + jn] = (uint32_t)words;
                    }
                    amax = max(amax, bm);
                    asm volatile("" : "+v"(amax));
                    if (jn == 1) cbase = co;
                }
            }
        };
        typedef std::true_type Y;
        typedef std::false_type NO;
        if constexpr (EPI != 0) {
            if (p.bits) { if (full) epi(Y(), Y()); else epi(Y(), NO()); }
            else { if (full) epi(NO(), Y()); else epi(NO(), NO()); }
        } else {
            if (full) epi(NO(), Y()); else epi(NO(), NO());
        }
        if (want_max) {   // (non-negative floats order like their bit patterns)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, o));
            if (lane == 0) atomic_max_bits(p.amax, amax);
        }
        if constexpr ((EPI & 2) != 0) __syncthreads();   // the next tile's prologue writes the LDS the shift values were read from
    }
    }   // tiles of this workgroup
}

// out = sum of the S split-K partials in fixed order (bit-reproducible), leaving max |out| like the product's own epilogue would
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ P, float* __restrict__ out, long n4, int S, unsigned* __restrict__ amax) {
    __shared__ float slots[4];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float am = 0.f;
    if (i < n4) {
        float4 a = reinterpret_cast<const float4*>(P)[i];
        for (int s_ = 1; s_ < S; ++s_) {
            const float4 v = reinterpret_cast<const float4*>(P)[(long)s_ * n4 + i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4*>(out)[i] = a;
        am = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
    }
    if (amax) block_max_bits(amax, wave_max(am), slots);
}

// A (M x K per batch, element (m, k) at A[b * a_sb + m * sm + k * sk]) -> the image.  Thread per 16-byte fragment slot; rows >= M and
// k >= K are zeros.
__global__ void split_a_kernel(const float* __restrict__ A, long a_sb, long sm, long sk, int nb, int M, int K, int rbp, int ktp, char* __restrict__ img) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)nb * ktp * rbp * 64;
    if (q >= total) return;
    const int lane = (int)(q & 63);
    long r = q >> 6;
    const int rb = (int)(r % rbp); r /= rbp;
    const int kt = (int)(r % ktp);
    const int b = (int)(r / ktp);
    const int m = rb * 32 + (lane & 31), k0 = kt * 16 + (lane >> 5) * 8;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (m < M && k0 + e < K) ? A[(long)b * a_sb + (long)m * sm + (long)(k0 + e) * sk] : 0.f;
    store_split8(x, img + gemm3_image_off(b, ktp, rbp, m, k0), rbp);
}

// the same for the f16x2 form: [batch][k-step][2 pieces][rbp][1 KB] of A 2^e, e from the bound *amax of |A| (one scale for all batches); thread 0
// records the inverse scale
__global__ void split_a_h2_kernel(const float* __restrict__ A, long a_sb, long sm, long sk, int nb, int M, int K, int rbp, int ktp, const unsigned* __restrict__ amax,
                                  char* __restrict__ img, float* __restrict__ inv_out) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = h2_exponent(*amax, 0);
    if (q == 0) inv_out[0] = h2_pow2(-e);
    const long total = (long)nb * ktp * rbp * 64;
    if (q >= total) return;
    const float s = h2_pow2(e);
    const int lane = (int)(q & 63);
    long r = q >> 6;
    const int rb = (int)(r % rbp); r /= rbp;
    const int kt = (int)(r % ktp);
    const int b = (int)(r / ktp);
    const int m = rb * 32 + (lane & 31), k0 = kt * 16 + (lane >> 5) * 8;
    uint32_t h[4], mm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = (m < M && k0 + 2 * i < K) ? A[(long)b * a_sb + (long)m * sm + (long)(k0 + 2 * i) * sk] : 0.f;
        const float x1 = (m < M && k0 + 2 * i + 1 < K) ? A[(long)b * a_sb + (long)m * sm + (long)(k0 + 2 * i + 1) * sk] : 0.f;
        split2_f16(x0 * s, x1 * s, h[i], mm[i]);
    }
    char* d = img + ((((long)b * ktp + kt) * 2) * rbp + rb) * 1024 + lane * 16;
    *reinterpret_cast<u32x4*>(d) = (u32x4){h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(d + (long)rbp * 1024) = (u32x4){mm[0], mm[1], mm[2], mm[3]};
}

// split_a_h2_kernel for a TABLE of filters in one launch (block -> task by binary search in the block prefix, as lgd_scale_rows_multi): the images
// of every trainable 1x1 convolution, W and W^T, once per step instead of a ~5 us launch in front of every product
__global__ __launch_bounds__(256) void split_a_h2_multi_kernel(const lgd_split_task* __restrict__ tasks, const int* __restrict__ blk0, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const lgd_split_task t = tasks[lo];
    const long q = (long)(blockIdx.x - blk0[lo]) * 256 + threadIdx.x;
    const int e = h2_exponent(*reinterpret_cast<const unsigned*>(t.amax), 0);
    if (q == 0) *reinterpret_cast<float*>(t.inv) = h2_pow2(-e);
    const int rbp = (t.M + 31) / 32, ktp = (t.K + 15) / 16;
    if (q >= (long)ktp * rbp * 64) return;
    const float s = h2_pow2(e);
    const float* A = reinterpret_cast<const float*>(t.a);
    const int lane = (int)(q & 63);
    const long r = q >> 6;
    const int rb = (int)(r % rbp), kt = (int)(r / rbp);
    const int m = rb * 32 + (lane & 31), k0 = kt * 16 + (lane >> 5) * 8;
    uint32_t h[4], mm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = (m < t.M && k0 + 2 * i < t.K) ? A[(long)m * t.sm + (long)(k0 + 2 * i) * t.sk] : 0.f;
        const float x1 = (m < t.M && k0 + 2 * i + 1 < t.K) ? A[(long)m * t.sm + (long)(k0 + 2 * i + 1) * t.sk] : 0.f;
        split2_f16(x0 * s, x1 * s, h[i], mm[i]);
    }
    char* d = reinterpret_cast<char*>(t.img) + (((long)kt * 2) * rbp + rb) * 1024 + lane * 16;
    *reinterpret_cast<u32x4*>(d) = (u32x4){h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(d + (long)rbp * 1024) = (u32x4){mm[0], mm[1], mm[2], mm[3]};
}

}  // namespace
}  // namespace lgd

extern "C" {

int lgd_gemm2h_split_multi(const void* tasks_dev, const int32_t* blk0_dev, int n, int nblocks, void* stream) {
    if (!tasks_dev || !blk0_dev || n < 1 || nblocks < 1) return LGD_EINVAL;
    LGD_LAUNCH("gemm2h_split_multi_kernel", lgd::split_a_h2_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
               reinterpret_cast<const lgd_split_task*>(tasks_dev), reinterpret_cast<const int*>(blk0_dev), n);
    return lgd::check_launch();
}

size_t lgd_gemm3_image_bytes(int nb, int M, int K) {
    if (nb <= 0 || M <= 0 || K <= 0) return 0;
    return (size_t)nb * ((K + 15) / 16) * 3 * ((M + 31) / 32) * 1024;
}

int lgd_gemm3_split(const float* A, long long a_sb, long long a_sm, long long a_sk, int nb, int M, int K, void* image, void* stream) {
    if (!A || !image || nb <= 0 || M <= 0 || K <= 0 || ((uintptr_t)image & 15)) return LGD_EINVAL;
    const int rbp = (M + 31) / 32, ktp = (K + 15) / 16;
    const long total = (long)nb * ktp * rbp * 64;
    LGD_LAUNCH("gemm3_split_kernel", lgd::split_a_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A, (long)a_sb,
               (long)a_sm, (long)a_sk, nb, M, K, rbp, ktp, (char*)image);
    return lgd::check_launch();
}

}  // extern "C"

template <int PCS>
static int gemm3_launch(const void* image, int image_shared, const float* a_inv, const float* B, const uint32_t* b_amax, long long b_sb, long long b_sk,
                        float* C, long long c_sb, long long c_sm, const float* R, long long r_sb, long long r_sm, const float* shift, int relu,
                        uint32_t* relu_bits, uint32_t* amax_out, int nb, int M, int N, int K, void* stream, float* splitk_ws = nullptr, int splits = 1) {
    if (!image || !B || !C || nb <= 0 || M <= 0 || N <= 0 || K <= 0 || (K & 15) || ((uintptr_t)image & 15)) return LGD_EINVAL;
    // 256-row tiles unless that leaves more than a quarter of the rows of the last tile empty and 128-row tiles do not
    const bool epi = R || shift || relu || relu_bits;
    // The f16x2 form (the student's 1x1 convolutions: HBM bound, 8 .. 130 flop per byte) always takes 128-row tiles: three workgroups per CU keep
    // more loads in flight than two, and a grid of twice as many tiles fills the chip where the 256-row one leaves CUs with one workgroup or none
    // (tools/gemm2h_probe.py, 8 images: res3 128 -> 512 + shortcut 188 -> 171 us, res4 256 -> 1024 + shortcut 144 -> 130, 1024 -> 256 97 -> 90;
    // B tiles are re-read by the neighbouring m-tiles from the same L2).  bf16x3 (the Winograd products: MFMA bound, B split once per 256 rows)
    // keeps the row-count rule.
    const bool small = PCS == 2 || ((M + 255) / 256 * 256 - M >= 64 && (M + 127) / 128 * 128 - M < 64);
    const int bm = small ? 128 : 256;
    lgd::Params p;
    p.rbp = (M + 31) / 32; p.ktp = K / 16;
    p.Aimg = (const char*)image; p.a_sb = image_shared ? 0 : (long)p.ktp * PCS * p.rbp * 1024;
    p.a_inv = a_inv; p.b_amax = b_amax;
    p.B = B; p.b_sb = (long)b_sb; p.b_ld = (long)b_sk;
    p.C = C; p.c_sb = (long)c_sb; p.c_ld = (long)c_sm;
    p.R = R; p.r_sb = (long)r_sb; p.r_ld = (long)r_sm; p.r_bytes = (((long)nb - 1) * r_sb + ((long)M - 1) * r_sm + N) * 4; p.shift = shift; p.bits = relu_bits; p.amax = amax_out; p.wpr = (N + 31) / 32; p.relu = relu ? 1 : 0;
    p.nb = nb; p.M = M; p.N = N; p.K = K; p.mt = (M + bm - 1) / bm; p.nt = (N + lgd::BN - 1) / lgd::BN;
    // split-K: S row blocks of the grid take K / S each and leave partials in the workspace [S][nb][M][N]; splitk_reduce_kernel adds them in fixed
    // order into C.  Plain products with a dense C only (an epilogue would have to move into the reduction).
    const int ksteps_all = K / 16;
    if (splits > 1) {
        if (!splitk_ws || epi || splits > ksteps_all || c_sm != N || c_sb != (long long)M * N || (((long long)nb * M * N) & 3)) return LGD_EINVAL;
        p.kper = (ksteps_all + splits - 1) / splits;
        splits = (ksteps_all + p.kper - 1) / p.kper;   // (no empty split)
        p.part_stride = (long)nb * M * N;
        p.C = splitk_ws;
        p.amax = nullptr;
    } else {
        splits = 1; p.kper = ksteps_all; p.part_stride = 0;
    }
    // 32-bit BYTE offsets from the descriptors' origins inside the kernel: one k-step of B rows, one tile of C / R rows
    if ((long)(lgd::BK + 1) * b_sk + N >= (1L << 30) || 256L * c_sm >= (1L << 30) || c_sm < 0 || (R && (r_sb < 0 || r_sm < 0 || 256L * r_sm >= (1L << 30))))
        return LGD_EINVAL;
    // 72 KB of dynamic LDS: above the default 64 KB limit.  hipFuncAttributeMaxDynamicSharedMemorySize and the CU count are per DEVICE: cached per
    // device index (a process that touched a second GPU launched without the attribute and with the first one's CU count)
    static bool attr_dev[64] = {};   // (one pair of caches per instantiation: PCS 3 / 2)
    static int cus_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LGD_ELAUNCH;
    bool& attr = attr_dev[dev];
    int& cus = cus_dev[dev];
    if constexpr (PCS == 3) if (!attr) {
        const void* big[5] = {(const void*)lgd::gemm3_kernel<256, 0, PCS>, (const void*)lgd::gemm3_kernel<256, 1, PCS>, (const void*)lgd::gemm3_kernel<256, 2, PCS>,
                              (const void*)lgd::gemm3_kernel<256, 3, PCS>, (const void*)lgd::gemm3_kernel<256, 4, PCS>};
        for (const void* f : big)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lgd::Tile<256, PCS>::LDS_BYTES) != hipSuccess) return LGD_ELAUNCH;
    }
    attr = true;
    p.total = (int)((image_shared ? ((long)nb * p.nt + 7) / 8 : (long)((nb + 7) / 8) * p.nt) * p.mt * 8);
    // persistent launch: as many workgroups as are resident at once (2 per CU with the 256-row tile, 3 with the 128-row one); each walks
    // tiles id, id + grid, ...  Measured equal to one workgroup per tile on every shape of tools/gemm3_probe.py (dispatch is not what
    // the kernel waits for), so the default stays one workgroup per tile (the hardware balances edge tiles); LGD_GEMM3_PERSIST=1 selects it
    if (!cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return LGD_ELAUNCH;
        cus = prop.multiProcessorCount;
    }
    const char* pe = getenv("LGD_GEMM3_PERSIST");
    const int slots = (cus * (small ? 3 : 2)) & ~7;
    const int gridn = (pe && pe[0] == '1') && p.total > slots ? slots : p.total;
    const dim3 grid((unsigned)(splits > 1 ? p.total : gridn), (unsigned)splits), block(lgd::NT);
    hipStream_t st = (hipStream_t)stream;
    const int kind = !epi ? 0 : (R ? 1 : 0) | (shift ? 2 : 0) ? (R ? 1 : 0) | (shift ? 2 : 0) : 4;
#define LGD_GEMM3_CASE(BM_, E_) \
    case E_: LGD_LAUNCH(PCS == 3 ? "gemm3_kernel" : "gemm2h_kernel", (lgd::gemm3_kernel<BM_, E_, PCS>), grid, block, (lgd::Tile<BM_, PCS>::LDS_BYTES), st, p); break;
    if (small) {
        switch (kind) { LGD_GEMM3_CASE(128, 0) LGD_GEMM3_CASE(128, 1) LGD_GEMM3_CASE(128, 2) LGD_GEMM3_CASE(128, 3) LGD_GEMM3_CASE(128, 4) }
    } else if constexpr (PCS == 3) {   // (the f16x2 form always takes the 128-row tile: its 256-row instances would be dead code)
        switch (kind) { LGD_GEMM3_CASE(256, 0) LGD_GEMM3_CASE(256, 1) LGD_GEMM3_CASE(256, 2) LGD_GEMM3_CASE(256, 3) LGD_GEMM3_CASE(256, 4) }
    }
#undef LGD_GEMM3_CASE
    if (splits > 1) {
        const long n4 = (long)nb * M * N / 4;
        LGD_LAUNCH("gemm3_splitk_reduce_kernel", lgd::splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, splitk_ws, C, n4, splits, amax_out);
    }
    return lgd::check_launch();
}

extern "C" {

int lgd_gemm3(const void* image, int image_shared, const float* B, long long b_sb, long long b_sk, float* C, long long c_sb, long long c_sm,
              const float* R, long long r_sb, long long r_sm, const float* shift, int relu, uint32_t* relu_bits, uint32_t* amax_out, int nb, int M, int N,
              int K, void* stream) {
    return gemm3_launch<3>(image, image_shared, nullptr, B, nullptr, b_sb, b_sk, C, c_sb, c_sm, R, r_sb, r_sm, shift, relu, relu_bits, amax_out, nb, M, N, K,
                           stream);
}

// ---- the f16x2 form of the same product (round 5): A as a two-piece f16 image scaled by a power of two (lgd_gemm2h_split: scale from the bound
// *a_amax of |A|, inverse recorded in a_inv), B split in registers after scaling by the power of two its bound *b_amax prescribes: three MFMAs per
// k-step instead of six and half the split arithmetic.  Same tiles, epilogues and arguments as lgd_gemm3.
size_t lgd_gemm2h_image_bytes(int nb, int M, int K) {
    if (nb <= 0 || M <= 0 || K <= 0) return 0;
    return (size_t)nb * ((K + 15) / 16) * 2 * ((M + 31) / 32) * 1024;
}

int lgd_gemm2h_split(const float* A, long long a_sb, long long a_sm, long long a_sk, int nb, int M, int K, const uint32_t* a_amax, void* image, float* a_inv,
                     void* stream) {
    if (!A || !image || !a_amax || !a_inv || nb <= 0 || M <= 0 || K <= 0 || ((uintptr_t)image & 15)) return LGD_EINVAL;
    const int rbp = (M + 31) / 32, ktp = (K + 15) / 16;
    const long total = (long)nb * ktp * rbp * 64;
    LGD_LAUNCH("gemm2h_split_kernel", lgd::split_a_h2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A, (long)a_sb,
               (long)a_sm, (long)a_sk, nb, M, K, rbp, ktp, a_amax, (char*)image, a_inv);
    return lgd::check_launch();
}

int lgd_gemm2h(const void* image, int image_shared, const float* a_inv, const float* B, const uint32_t* b_amax, long long b_sb, long long b_sk, float* C,
               long long c_sb, long long c_sm, const float* R, long long r_sb, long long r_sm, const float* shift, int relu, uint32_t* relu_bits,
               uint32_t* amax_out, float* splitk_ws, int splits, int nb, int M, int N, int K, void* stream) {
    if (!a_inv || !b_amax || !image_shared) return LGD_EINVAL;   // (one image and one scale for all batches: the student's 1x1 convolutions)
    return gemm3_launch<2>(image, image_shared, a_inv, B, b_amax, b_sb, b_sk, C, c_sb, c_sm, R, r_sb, r_sm, shift, relu, relu_bits, amax_out, nb, M, N, K,
                           stream, splitk_ws, splits);
}

}  // extern "C"
