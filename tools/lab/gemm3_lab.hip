// LAB: strided-batched fp32 GEMM on the bf16 MFMA pipe with three-way split operands.
//   C[b] (M x N) = A[b] (M x K) . B[b] (K x N),  fp32 in HBM on both sides and in the result.
// Every fp32 operand element x is split while it is STAGED into LDS:  x = h + m + l,  h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
// (round to nearest: |x - h - m - l| <= 2^-24 |x| up to bf16 underflow), and 6 of the 9 cross products are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16:   a.b ~= ah.bh + (ah.bm + am.bh) + (ah.bl + am.bm + al.bh)      (dropped: am.bl + al.bm + al.bl <= ~2^-23 |a||b|)
// The frequency buffers stay fp32 [C][64][T] (no extra HBM bytes, the Winograd transforms are untouched); the split costs ~5.5 VALU
// operations per staged element, issued beside the MFMAs.
//
// Tile: 128 x 128 x 32 per 256-thread workgroup (2 x 2 waves, 64 x 64 per wave = 2 x 2 MFMA blocks of 32 x 32), one LDS buffer of
// 48 KB (3 pieces x (128 + 128) rows x 32 k x 2 B) so that three workgroups share a CU and one's staging hides under the others' MFMAs;
// the next k-tile's global loads are in flight (registers) during the MFMA phase.
// LDS image: per piece and 32-row block, per 16-deep k-step, the 64 lanes' 16-byte fragments in lane order (a wave's ds_read_b128 covers
// 1 KB linearly), the 16-byte slot XOR-ed with the (k-group, k-step) bits so that the staging stores are conflict-free as well.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int PIECE_BYTES = 128 * BK * 2;          // one piece of one operand: 8 KB
constexpr int OPER_BYTES = 3 * PIECE_BYTES;        // 24 KB
constexpr int LDS_BYTES = 2 * OPER_BYTES;          // 48 KB

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const bf16x2 v = __builtin_convertvector((f32x2){a, b}, bf16x2);   // v_cvt_pk_bf16_f32 (RNE), a in the low half
    return __builtin_bit_cast(uint32_t, v);
}
// two floats -> three packed bf16 pairs
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = pack_bf16(x0, x1);
    float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = pack_bf16(r0, r1);
    r0 -= __builtin_bit_cast(float, m << 16);
    r1 -= __builtin_bit_cast(float, m & 0xffff0000u);
    l = pack_bf16(r0, r1);
}
// byte offset of the 16-byte fragment (row r of 128, 8-deep k-group kg of 4) inside one piece
__device__ __forceinline__ int frag_off(int r, int kg) {
    const int rb = r >> 5, rr = r & 31, ks = kg >> 1, g = kg & 1;
    return ((rb * 2 + ks) << 10) + (g << 9) + ((rr << 4) ^ (g << 6) ^ (ks << 5));
}

// ---- staging, operand whose k axis is contiguous in memory: X(row, k) at X[row * ld + k].  Thread <-> (row, k-group) pairs q = t, t + 256.
struct StageK {
    float4 v[2][2];
    __device__ __forceinline__ void load(const float* __restrict__ X, long ld, int row0, int rows, int k0, int K, int t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = t + NT * j, r = q >> 2, kg = q & 3;
            const int row = row0 + r, k = k0 + kg * 8;
            const bool ok = row < rows && k < K;            // K % 8 == 0 (host-checked): a k-group is all in or all out
            const float* p = X + (long)(ok ? row : 0) * ld + (ok ? k : 0);
            // (no select on the loaded value HERE: it would put the wait for the load in front of the MFMA phase it is meant to overlap)
            v[j][0] = *reinterpret_cast<const float4*>(p);
            v[j][1] = *reinterpret_cast<const float4*>(p + 4);
        }
    }
    __device__ __forceinline__ void store(char* lds, int row0, int rows, int k0, int K, int t) const {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = t + NT * j, r = q >> 2, kg = q & 3;
            const float z = (row0 + r < rows && k0 + kg * 8 < K) ? 1.f : 0.f;   // out-of-range elements enter the product as zeros
            uint32_t h[4], m[4], l[4];
            split2(v[j][0].x * z, v[j][0].y * z, h[0], m[0], l[0]);
            split2(v[j][0].z * z, v[j][0].w * z, h[1], m[1], l[1]);
            split2(v[j][1].x * z, v[j][1].y * z, h[2], m[2], l[2]);
            split2(v[j][1].z * z, v[j][1].w * z, h[3], m[3], l[3]);
            const int o = frag_off(r, kg);
            *reinterpret_cast<u32x4*>(lds + o) = (u32x4){h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4*>(lds + PIECE_BYTES + o) = (u32x4){m[0], m[1], m[2], m[3]};
            *reinterpret_cast<u32x4*>(lds + 2 * PIECE_BYTES + o) = (u32x4){l[0], l[1], l[2], l[3]};
        }
    }
};
// ---- staging, operand whose row (m or n) axis is contiguous: X(row, k) at X[k * ld + row].  Thread <-> (k-group = wave, rows 2*lane, 2*lane+1).
struct StageR {
    float2 v[8];
    __device__ __forceinline__ void load(const float* __restrict__ X, long ld, int row0, int rows, int k0, int K, int t) {
        const int kg = t >> 6, r = (t & 63) * 2;
        const int row = row0 + r;
        const bool rok = row < rows;                        // rows % 2 == 0 (host-checked)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + kg * 8 + e;
            const bool ok = rok && k < K;
            const float* p = X + (long)(ok ? k : 0) * ld + (ok ? row : 0);
            v[e] = *reinterpret_cast<const float2*>(p);
        }
    }
    __device__ __forceinline__ void store(char* lds, int row0, int rows, int k0, int K, int t) const {
        const int kg = t >> 6, r = (t & 63) * 2;
        const bool rok = row0 + r < rows;
        uint32_t h0[4], m0[4], l0[4], h1[4], m1[4], l1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float za = (rok && k0 + kg * 8 + 2 * e < K) ? 1.f : 0.f, zb = (rok && k0 + kg * 8 + 2 * e + 1 < K) ? 1.f : 0.f;
            split2(v[2 * e].x * za, v[2 * e + 1].x * zb, h0[e], m0[e], l0[e]);
            split2(v[2 * e].y * za, v[2 * e + 1].y * zb, h1[e], m1[e], l1[e]);
        }
        const int o0 = frag_off(r, kg), o1 = frag_off(r + 1, kg);
        *reinterpret_cast<u32x4*>(lds + o0) = (u32x4){h0[0], h0[1], h0[2], h0[3]};
        *reinterpret_cast<u32x4*>(lds + o1) = (u32x4){h1[0], h1[1], h1[2], h1[3]};
        *reinterpret_cast<u32x4*>(lds + PIECE_BYTES + o0) = (u32x4){m0[0], m0[1], m0[2], m0[3]};
        *reinterpret_cast<u32x4*>(lds + PIECE_BYTES + o1) = (u32x4){m1[0], m1[1], m1[2], m1[3]};
        *reinterpret_cast<u32x4*>(lds + 2 * PIECE_BYTES + o0) = (u32x4){l0[0], l0[1], l0[2], l0[3]};
        *reinterpret_cast<u32x4*>(lds + 2 * PIECE_BYTES + o1) = (u32x4){l1[0], l1[1], l1[2], l1[3]};
    }
};

struct Params {
    const float* A; long a_sb, a_ld;      // A(m, k): AK ? A[m * a_ld + k] : A[k * a_ld + m]
    const float* B; long b_sb, b_ld;      // B(k, n): BK_ ? B[n * b_ld + k] : B[k * b_ld + n]
    float* C; long c_sb, c_ld;            // C(m, n) at C[m * c_ld + n]
    int nb, M, N, K, mt, nt;              // mt / nt: tiles along M / N
    int ksplit; long c_ss;                // split-K: slice s of batch b writes C + s * c_ss (partials summed by the caller)
};

template <bool AK, bool BKC>
__global__ __launch_bounds__(NT) void gemm3_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
    // workgroup -> (batch, n-tile, m-tile, k-slice): ids that differ by 8 share an XCD (round-robin dispatch), so the m-tiles that
    // read the same B tile, then the n-tiles of one batch (same A), are neighbours on ONE XCD's L2
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int per = p.mt * p.ksplit;
    const int sub = j % per, rest = (j / per) * 8 + xcd;
    if (rest >= p.nb * p.nt) return;
    const int tm = sub % p.mt, ksl = sub / p.mt;
    const int b = rest / p.nt, tn = rest % p.nt;
    const int m0 = tm * BM, n0 = tn * BN;
    const int ktiles = (p.K + BK - 1) / BK;
    const int kt0 = (int)((long)ktiles * ksl / p.ksplit), kt1 = (int)((long)ktiles * (ksl + 1) / p.ksplit);
    const float* A = p.A + (long)b * p.a_sb;
    const float* B = p.B + (long)b * p.b_sb;
    char* ldsA = lds;
    char* ldsB = lds + OPER_BYTES;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    typename std::conditional<AK, StageK, StageR>::type sa;
    typename std::conditional<BKC, StageK, StageR>::type sb;
    if (kt0 < kt1) {
        sa.load(A, p.a_ld, m0, p.M, kt0 * BK, p.K, t);
        sb.load(B, p.b_ld, n0, p.N, kt0 * BK, p.K, t);
    }
    // this lane's fragment slot inside a (32-row block, k-step) sub-block: row lane & 31, k-group lane >> 5
    const int g = lane >> 5, rr = lane & 31;
    for (int kt = kt0; kt < kt1; ++kt) {
        sa.store(ldsA, m0, p.M, kt * BK, p.K, t);
        sb.store(ldsB, n0, p.N, kt * BK, p.K, t);
        __syncthreads();
        if (kt + 1 < kt1) {
            sa.load(A, p.a_ld, m0, p.M, (kt + 1) * BK, p.K, t);
            sb.load(B, p.b_ld, n0, p.N, (kt + 1) * BK, p.K, t);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = (g << 9) + ((rr << 4) ^ (g << 6) ^ (ks << 5));
            bf16x8 fb[3][2];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
                    fb[pc][jn] = *reinterpret_cast<const bf16x8*>(ldsB + pc * PIECE_BYTES + (((wn * 2 + jn) * 2 + ks) << 10) + slot);
#pragma unroll
            for (int pa = 2; pa >= 0; --pa) {            // smallest pieces first
                bf16x8 fa[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fa[i] = *reinterpret_cast<const bf16x8*>(ldsA + pa * PIECE_BYTES + (((wm * 2 + i) * 2 + ks) << 10) + slot);
#pragma unroll
                for (int pb = 2 - pa; pb >= 0; --pb)     // pa + pb <= 2: the six kept products
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn)
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[pb][jn], acc[i][jn], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); 32-bit offsets from one base
    const int mw = m0 + wm * 64 + 4 * g, nw = n0 + wn * 64 + rr;
    float* C = p.C + (long)b * p.c_sb + (long)ksl * p.c_ss + (long)mw * p.c_ld + nw;
    const int ld = (int)p.c_ld;
    if (m0 + BM <= p.M && n0 + BN <= p.N) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) C[(i * 32 + (e & 3) + 8 * (e >> 2)) * ld + jn * 32] = acc[i][jn][e];
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int dm = i * 32 + (e & 3) + 8 * (e >> 2);
                    if (mw + dm < p.M && nw + jn * 32 < p.N) C[dm * ld + jn * 32] = acc[i][jn][e];
                }
    }
}


// ======================================================================================================================================
// v3: the A operand arrives PRE-SPLIT (it is a filter: small, reused by every n-tile, changes once per step), as a bf16 image in the
// exact order the MFMA fragments are read -- [batch][k-step of 16][piece][32-row block][lane][8 bf16] -- so that a k-step's share of a
// 256-row tile is 3 runs of 8 KB which LDS-DMA (global_load_lds_dwordx4: no registers, no VALU) drops into LDS unchanged.  Only B (the
// activations, each element staged exactly once in the whole product because the tile spans all 256 rows of A) is split in the kernel:
// ~1.3 VALU instructions per MFMA instead of ~8.
// Tile 256 x 128 x 16, 256 threads (2 x 2 waves, 128 x 64 per wave), two LDS buffers of 36 KB: one barrier per k-step, two workgroups
// per CU.
#ifndef XMAP
#define XMAP 1
#endif
namespace s3 {
constexpr int BM = 256, BN = 128, BK = 16, NT = 256;
constexpr int A_BYTES = 3 * 8 * 1024, B_BYTES = 3 * 4 * 1024, BUF = A_BYTES + B_BYTES, LDS_BYTES = 2 * BUF;   // 72 KB

#define GLDS16(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

struct Params {
    const char* Aimg; long a_sb; int rbp, ktp;   // image: [nb][ktp][3][rbp][1024 B]; a_sb in bytes
    const float* B; long b_sb, b_ld;             // B(k, n) at B[k * b_ld + n]
    float* C; long c_sb, c_ld;
    int nb, M, N, K, mt, nt;
};

// (A hand-counted variant -- every load of the k-loop an asm statement, B two k-steps ahead in two register sets, counted vmcnt(14) /
//  vmcnt(8) so that the DMA wait leaves the B loads in flight -- measured 281 us against 285 us for this compiler-tracked form and
//  produced wrong tiles under load: hipcc may copy an asm load's destination register at the loop back-edge before the data lands
//  (guide 5.7 item 1).  Not kept.)
#ifndef ABL
#define ABL 0     // lab ablations: 1 no staging in the loop, 2 no staging + no barriers, 3 no MFMA, 4 no epilogue stores, 5 no split + ds_write of B, 6 no image DMA, 7 no B loads
#endif
__global__ __launch_bounds__(NT) void gemm3s_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 1, wn = w & 1;
    // workgroup -> (batch, n-tile, m-tile).  Consecutive ids go round-robin to the 8 XCDs: XCD x takes the batches b = x (mod 8) and walks
    // their tiles in order, so that its L2 holds the images of the one or two batches it is working on (XMAP 0: tiles of one batch
    // spread over all XCDs, every L2 holding all ~13 images in flight)
#ifndef XMAP
#define XMAP 1
#endif
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
#if XMAP
    const int per_b = p.nt * p.mt;
    const int b = (j / per_b) * 8 + xcd, r = j % per_b;
    if (b >= p.nb) return;
    const int tn = r / p.mt, sub = r % p.mt;
#else
    const int sub = j % p.mt, rest = (j / p.mt) * 8 + xcd;
    if (rest >= p.nb * p.nt) return;
    const int b = rest / p.nt, tn = rest % p.nt;
#endif
    const int m0 = sub * BM, n0 = tn * BN, rb0 = sub * 8;
    const int ksteps = p.K / BK;                       // K % 16 == 0 (host-checked)
    const char* Ai = p.Aimg + (long)b * p.a_sb;
    // B staging: thread <-> (k-group kg = t >> 7, column n = t & 127): 8 dwords down the k axis, a wave's load covers 256 contiguous bytes.
    // Addresses are a uniform per-k-step base (scalar registers) + eight per-thread 32-bit offsets computed ONCE: no address arithmetic
    // in the loop.  Columns >= N are read from column N - 1 and rows >= M come as zeros from the image: both only reach elements of C
    // that are never stored (rows and columns of a product are independent), so nothing is zeroed here.
    const int kg = t >> 7, nl = t & 127;
    const int ncol = n0 + nl < p.N ? n0 + nl : p.N - 1;
    const float* Bb = p.B + (long)b * p.b_sb;
    uint32_t boff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) boff[e] = (uint32_t)((kg * 8 + e) * (int)p.b_ld + ncol);
    const long bstep = (long)BK * p.b_ld;
    const int bslot = A_BYTES + (nl >> 5) * 1024 + kg * 512 + (nl & 31) * 16;
    float bv[8];
    auto load_b = [&](int ks) {
        const float* Bk = Bb + ks * bstep;
#pragma unroll
#ifdef NTB
        for (int e = 0; e < 8; ++e) bv[e] = __builtin_nontemporal_load(Bk + boff[e]);
#else
        for (int e = 0; e < 8; ++e) bv[e] = Bk[boff[e]];
#endif
    };
    auto store_b = [&](char* buf) {
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2(bv[2 * e], bv[2 * e + 1], h[e], m[e], l[e]);
        char* d = buf + bslot;
        *reinterpret_cast<u32x4*>(d) = (u32x4){h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(d + 4096) = (u32x4){m[0], m[1], m[2], m[3]};
        *reinterpret_cast<u32x4*>(d + 8192) = (u32x4){l[0], l[1], l[2], l[3]};
    };
    // A: 24 chunks of 1 KB per k-step ([piece][row block]); wave w moves chunks 6w .. 6w+5: per-thread 32-bit offsets once, the k-step's
    // base is uniform
    uint32_t aoff[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int ch = w * 6 + c, pc = ch >> 3, rbl = ch & 7;
        int rb = rb0 + rbl;
        rb = rb < p.rbp ? rb : p.rbp - 1;
        aoff[c] = (uint32_t)((pc * p.rbp + rb) * 1024 + lane * 16);
    }
    const long astep = (long)3 * p.rbp * 1024;
    auto dma_a = [&](int ks, char* buf) {
        const char* Ak = Ai + ks * astep;
#pragma unroll
        for (int c = 0; c < 6; ++c) GLDS16(Ak + aoff[c], buf + (w * 6 + c) * 1024);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // prologue: k-step 0 into buffer 0, B of k-step 1 into registers
    dma_a(0, lds);
    load_b(0);
    store_b(lds);
    if (ksteps > 1) load_b(1);
    __syncthreads();
    const int slot = lane * 16;
#if ABL == 8 || ABL == 9
    // variant: the DMA is issued first, the split of B(ks+1) sits INSIDE the MFMA phase (after the first 8 MFMAs; ABL 9: paired 1:1 with
    // MFMAs by sched_group_barrier), its three LDS stores and the loads of B(ks+2) follow the last MFMA
    for (int ks = 0; ks < ksteps; ++ks) {
        char* cur = lds + (ks & 1) * BUF;
        char* nxt = lds + ((ks + 1) & 1) * BUF;
        // one basic block per k-step (nothing conditional: past the end the DMA and the loads repeat the last k-step into the unused buffer),
        // so that the scheduler may place the split's VALU work between the MFMAs
        const int k1 = ks + 1 < ksteps ? ks + 1 : ksteps - 1, k2 = ks + 2 < ksteps ? ks + 2 : ksteps - 1;
        dma_a(k1, nxt);
        bf16x8 fb[3][2];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
                fb[pc][jn] = *reinterpret_cast<const bf16x8*>(cur + A_BYTES + pc * 4096 + (wn * 2 + jn) * 1024 + slot);
        uint32_t sh[4], sm[4], sl[4];
#pragma unroll
        for (int pa = 2; pa >= 0; --pa) {
            bf16x8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(cur + pa * 8192 + (wm * 4 + i) * 1024 + slot);
#pragma unroll
            for (int pb = 2 - pa; pb >= 0; --pb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[pb][jn], acc[i][jn], 0, 0, 0);
            if (pa == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) split2(bv[2 * e], bv[2 * e + 1], sh[e], sm[e], sl[e]);
            }
        }
#if ABL == 9
        // 48 MFMAs, ~40 VALU: one VALU behind each MFMA
#pragma unroll
        for (int q = 0; q < 40; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
#endif
        {
            char* d = nxt + bslot;
            *reinterpret_cast<u32x4*>(d) = (u32x4){sh[0], sh[1], sh[2], sh[3]};
            *reinterpret_cast<u32x4*>(d + 4096) = (u32x4){sm[0], sm[1], sm[2], sm[3]};
            *reinterpret_cast<u32x4*>(d + 8192) = (u32x4){sl[0], sl[1], sl[2], sl[3]};
            load_b(k2);
        }
        __syncthreads();
    }
#else
    for (int ks = 0; ks < ksteps; ++ks) {
        char* cur = lds + (ks & 1) * BUF;
        char* nxt = lds + ((ks + 1) & 1) * BUF;
#if ABL != 1 && ABL != 2
        if (ks + 1 < ksteps) {
            // order matters to hipcc's wait insertion: the use of bv (loaded a whole k-step ago) comes BEFORE the LDS-DMA is issued --
            // with a DMA in flight the compiler waits vmcnt(0) at the next use of an ordinary load's result, which would expose the DMA
#if ABL != 5
            store_b(nxt);
#endif
#if ABL != 6
            dma_a(ks + 1, nxt);
#endif
#if ABL != 7
            if (ks + 2 < ksteps) load_b(ks + 2);
#else
            asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]));
#endif
        }
#endif
        bf16x8 fb[3][2];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
                fb[pc][jn] = *reinterpret_cast<const bf16x8*>(cur + A_BYTES + pc * 4096 + (wn * 2 + jn) * 1024 + slot);
#pragma unroll
        for (int pa = 2; pa >= 0; --pa) {
            bf16x8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(cur + pa * 8192 + (wm * 4 + i) * 1024 + slot);
#if ABL == 3
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[i]));
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) asm volatile("" :: "v"(fb[pa][jn]));
#else
#pragma unroll
            for (int pb = 2 - pa; pb >= 0; --pb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[pb][jn], acc[i][jn], 0, 0, 0);
#endif
        }
#if ABL != 2
        __syncthreads();
#endif
    }
#endif
    const int g = lane >> 5, rr = lane & 31;
    const int mw = m0 + wm * 128 + 4 * g, nw = n0 + wn * 64 + rr;
    float* C = p.C + (long)b * p.c_sb + (long)mw * p.c_ld + nw;
    const int ld = (int)p.c_ld;
#if ABL == 4
    if (ksteps == 12345)
#endif
    if (m0 + BM <= p.M && n0 + BN <= p.N) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) C[(i * 32 + (e & 3) + 8 * (e >> 2)) * ld + jn * 32] = acc[i][jn][e];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int dm = i * 32 + (e & 3) + 8 * (e >> 2);
                    if (mw + dm < p.M && nw + jn * 32 < p.N) C[dm * ld + jn * 32] = acc[i][jn][e];
                }
    }
}

// A (M x K per batch, element (m, k) at A[m * sm + k * sk]) -> the image above.  Thread per 16-byte fragment slot.
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
    uint32_t h[4], mm[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2(x[2 * e], x[2 * e + 1], h[e], mm[e], l[e]);
    char* d = img + ((((long)b * ktp + kt) * 3) * rbp + rb) * 1024 + lane * 16;
    *reinterpret_cast<u32x4*>(d) = (u32x4){h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(d + (long)rbp * 1024) = (u32x4){mm[0], mm[1], mm[2], mm[3]};
    *reinterpret_cast<u32x4*>(d + 2 * (long)rbp * 1024) = (u32x4){l[0], l[1], l[2], l[3]};
}
}  // namespace s3

}  // namespace

// A(m,k): a_sk == 1 (k contiguous, a_sm = leading dimension) or a_sm == 1 (m contiguous, a_sk = leading dimension); B(k,n) alike;
// C(m,n) with n contiguous.  Strides in elements.  ksplit > 1: slice s writes its partial product to C + s * c_ss.
extern "C" int gemm3_lab(const float* A, long a_sb, long a_sm, long a_sk, const float* B, long b_sb, long b_sk, long b_sn,
                         float* C, long c_sb, long c_sm, int nb, int M, int N, int K, int ksplit, long c_ss, void* stream) {
    const bool ak = a_sk == 1, bk = b_sk == 1 && b_sn != 1;
    if (!ak && a_sm != 1) return -1;
    if (!bk && b_sn != 1) return -1;
    Params p;
    p.A = A; p.a_sb = a_sb; p.a_ld = ak ? a_sm : a_sk;
    p.B = B; p.b_sb = b_sb; p.b_ld = bk ? b_sn : b_sk;
    p.C = C; p.c_sb = c_sb; p.c_ld = c_sm;
    p.nb = nb; p.M = M; p.N = N; p.K = K; p.mt = (M + BM - 1) / BM; p.nt = (N + BN - 1) / BN;
    p.ksplit = ksplit < 1 ? 1 : ksplit; p.c_ss = c_ss;
    // alignment the vector loads need
    if (ak && ((K & 7) || (p.a_ld & 3) || (a_sb & 3) || ((uintptr_t)A & 15))) return -2;
    if (!ak && ((M & 1) || (p.a_ld & 1) || (a_sb & 1) || ((uintptr_t)A & 7))) return -2;
    if (bk && ((K & 7) || (p.b_ld & 3) || (b_sb & 3) || ((uintptr_t)B & 15))) return -2;
    if (!bk && ((N & 1) || (p.b_ld & 1) || (b_sb & 1) || ((uintptr_t)B & 7))) return -2;
    const long groups = ((long)nb * p.nt + 7) / 8;
    const long grid = groups * 8 * p.mt * p.ksplit;
    hipStream_t s = (hipStream_t)stream;
    if (ak && !bk) hipLaunchKernelGGL((gemm3_kernel<true, false>), dim3((unsigned)grid), dim3(NT), LDS_BYTES, s, p);
    else if (!ak && !bk) hipLaunchKernelGGL((gemm3_kernel<false, false>), dim3((unsigned)grid), dim3(NT), LDS_BYTES, s, p);
    else if (ak && bk) hipLaunchKernelGGL((gemm3_kernel<true, true>), dim3((unsigned)grid), dim3(NT), LDS_BYTES, s, p);
    else hipLaunchKernelGGL((gemm3_kernel<false, true>), dim3((unsigned)grid), dim3(NT), LDS_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// bytes of the pre-split image of an (nb, M, K) operand
extern "C" long gemm3_image_bytes(int nb, int M, int K) { return (long)nb * ((K + 15) / 16) * 3 * ((M + 31) / 32) * 1024; }
// A(m, k) at A[b * a_sb + m * a_sm + k * a_sk] -> image
extern "C" int gemm3_split_a(const float* A, long a_sb, long a_sm, long a_sk, int nb, int M, int K, void* img, void* stream) {
    const int rbp = (M + 31) / 32, ktp = (K + 15) / 16;
    const long total = (long)nb * ktp * rbp * 64;
    hipLaunchKernelGGL(s3::split_a_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A, a_sb, a_sm, a_sk, nb, M, K, rbp, ktp,
                       (char*)img);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
// C[b] = A[b] . B[b] with A given as its image; B(k, n) with n contiguous (b_sn == 1), C(m, n) with n contiguous
extern "C" int gemm3s_lab(const void* Aimg, const float* B, long b_sb, long b_sk, float* C, long c_sb, long c_sm, int nb, int M, int N, int K, void* stream) {
    s3::Params p;
    p.rbp = (M + 31) / 32; p.ktp = (K + 15) / 16;
    p.Aimg = (const char*)Aimg; p.a_sb = (long)p.ktp * 3 * p.rbp * 1024;
    p.B = B; p.b_sb = b_sb; p.b_ld = b_sk;
    p.C = C; p.c_sb = c_sb; p.c_ld = c_sm;
    p.nb = nb; p.M = M; p.N = N; p.K = K; p.mt = (M + s3::BM - 1) / s3::BM; p.nt = (N + s3::BN - 1) / s3::BN;
    if ((K & 15) || (long)(K + 16) * b_sk >= (1L << 31) || (long)p.ktp * 3 * p.rbp * 1024 >= (1L << 31)) return -2;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)s3::gemm3s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, s3::LDS_BYTES); attr = true; }
#if XMAP
    const long groups = (long)((nb + 7) / 8) * p.nt;
#else
    const long groups = ((long)nb * p.nt + 7) / 8;
#endif
    hipLaunchKernelGGL(s3::gemm3s_kernel, dim3((unsigned)(groups * 8 * p.mt)), dim3(s3::NT), s3::LDS_BYTES, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
