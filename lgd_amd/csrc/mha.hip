// K2: inter-object relation adaptation = nn.MultiheadAttention(E=256, heads=8) between appearance and
// label embeddings with a block-diagonal (per-image) mask  [ref: dynamic_teacher.py:76-78, 255-273].
//
// The reference calls the module once per FPN level (5x), each call re-projecting the shared K/V
// operand and launching ~8 tiny kernels over a (T,T) masked score matrix.  Here:
//   * lgd_gemm_batch : fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32 fma chains) for the dense in/out
//     projections of ALL levels in one launch (a list of strided GEMM problems; also serves the
//     backward's dX / dW / dbias products);
//   * mha core       : one wave64 per (level, image, head); lanes own query rows (forward, dQ) or key rows
//     (dK, dV), the partner rows are broadcast from LDS, softmax is an online max/sum per lane --
//     only the image's own n x n block is ever touched (no (T,T) mask); the backward emits per-level
//     partial gradients, a shared operand's gradient is their fixed-order sum (no atomics).
// Sizes are tiny (T ~ 10^2 tokens): these kernels are latency-bound, not roofline-bound; MFMA is used
// because the projections are GEMM-shaped, not because it pays.
#include "common.h"

namespace lgd {

// ------------------------------------------------------------------------------------------- GEMM list
constexpr int kMaxProb = 16;
struct GemmProb {
    const float* A; const float* B; const float* bias; float* C; float* rowsum;
    int M, N, K;
    long long sa_m, sa_k, sb_n, sb_k, sc_m, sc_n;
    float alpha;
    int tile0, tiles_n;
    int vec;   // both operands qualify for the branch-free 16-byte fetch (gemm_fetch_vec)
};
struct GemmArgs { GemmProb p[kMaxProb]; int np, ntiles; };

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kBM = 64, kBN = 64, kBK = 64;   // block tile; 4 waves, wave w owns rows [16w, 16w+16) x all 64 columns
constexpr int kLd = kBM + 1;                   // LDS row stride (floats): +1 breaks the 4-way write conflict

// C[m,n] = alpha * (sum_k A(m,k) * B(n,k) + bias[n]);  rowsum[m] = alpha * sum_k A(m,k) (optional, n-tile 0 only)
// Each operand must be contiguous along m (n) or along k.  Per k-chunk of 64 every thread fetches four 16-byte
// vectors of A and four of B along the operand's contiguous axis (coalesced), the chunk is transposed into LDS as
// [k][m] / [k][n], and each wave issues 64 v_mfma_f32_16x16x4_f32 (A: lane l holds A[i=l&15][k=l>>4];
// B: B[k=l>>4][j=l&15]; C/D: col=l&15, row=(l>>4)*4+reg).  The next chunk's global loads are in flight while
// the current one is multiplied (register-staged double buffering).
__device__ __forceinline__ float4 gemm_fetch(const float* base, long long s_row, long long s_k, int row0, int nrows, int k0, int K, int tid,
                                             int& r, int& kk) {
    // returns 4 consecutive elements along the contiguous axis; (r, kk) = tile coordinates of element 0
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s_k == 1) {            // k-contiguous: 4 threads per row
        r = tid >> 2; kk = (tid & 3) * 4;
        if (row0 + r < nrows) {
            const float* p = base + (long long)(row0 + r) * s_row + (k0 + kk);
            if (k0 + kk + 3 < K && ((reinterpret_cast<size_t>(p) & 15) == 0)) v = *reinterpret_cast<const float4*>(p);
            else {
                if (k0 + kk + 0 < K) v.x = p[0];
                if (k0 + kk + 1 < K) v.y = p[1];
                if (k0 + kk + 2 < K) v.z = p[2];
                if (k0 + kk + 3 < K) v.w = p[3];
            }
        }
    } else {                   // row-contiguous: 16 threads per k
        kk = tid >> 4; r = (tid & 15) * 4;
        if (k0 + kk < K) {
            const float* p = base + (long long)(k0 + kk) * s_k + (row0 + r);
            if (row0 + r + 3 < nrows && ((reinterpret_cast<size_t>(p) & 15) == 0)) v = *reinterpret_cast<const float4*>(p);
            else {
                if (row0 + r + 0 < nrows) v.x = p[0];
                if (row0 + r + 1 < nrows) v.y = p[1];
                if (row0 + r + 2 < nrows) v.z = p[2];
                if (row0 + r + 3 < nrows) v.w = p[3];
            }
        }
    }
    return v;
}
__device__ __forceinline__ void gemm_stage(float* lds, const float4& v, bool k_contig, int r, int kk) {
    if (k_contig) { lds[(kk + 0) * kLd + r] = v.x; lds[(kk + 1) * kLd + r] = v.y; lds[(kk + 2) * kLd + r] = v.z; lds[(kk + 3) * kLd + r] = v.w; }
    else { lds[kk * kLd + r + 0] = v.x; lds[kk * kLd + r + 1] = v.y; lds[kk * kLd + r + 2] = v.z; lds[kk * kLd + r + 3] = v.w; }
}

// Vectorised operand fetch without control flow: every access is an aligned 16-byte load at an address clamped into the operand,
// elements beyond M / N / K are zeroed by a select.  Straight-line loads let the compiler COUNT the waits (s_waitcnt vmcnt(n)), which
// is what keeps the loads of chunk c+2 in flight while chunk c+1 is staged (the branchy form below forces vmcnt(0) at every join).
// Needs the operand's contiguous extent and its other stride to be multiples of 4 elements and a 16-byte aligned base (p.vec).
__device__ __forceinline__ float4 gemm_fetch_vec(const float* base, long long s_row, long long s_k, int row0, int nrows, int k0, int K, int tid,
                                                 int& r, int& kk, bool& ok) {
    const float* p;
    if (s_k == 1) {            // k-contiguous: 4 threads per row
        r = tid >> 2; kk = (tid & 3) * 4;
        ok = row0 + r < nrows && k0 + kk < K;
        p = base + (long long)min(row0 + r, nrows - 1) * s_row + min(k0 + kk, K - 4);
    } else {                   // row-contiguous: 16 threads per k
        kk = tid >> 4; r = (tid & 15) * 4;
        ok = k0 + kk < K && row0 + r < nrows;
        p = base + (long long)min(k0 + kk, K - 1) * s_k + min(row0 + r, nrows - 4);
    }
    return *reinterpret_cast<const float4*>(p);   // the caller zeroes it where !ok -- at STAGING time, so that nothing waits on the load here
}

// Software pipeline over the k-chunks, prefetch distance TWO: while chunk c is multiplied out of LDS buffer c&1, chunk c+1 sits in one
// register set (staged into the other LDS buffer after the MFMAs) and the loads of chunk c+2 are issued into the other set.  These
// GEMMs have 10^2 rows -- a handful of workgroups, each a serial chain of K/64 chunks -- so a launch costs (chunks x exposed latency):
// with distance one (round 1) every chunk exposed a full memory latency, 5.5 us per chunk (89 us for K = 1024).
template <bool VEC>
__device__ __forceinline__ void gemm_tile(const GemmProb& p, int m0, int n0, float (*sA)[kBK * kLd], float (*sB)[kBK * kLd]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const bool a_kc = p.sa_k == 1, b_kc = p.sb_k == 1;
    const long long sa_row = a_kc ? p.sa_m : 1, sa_kk = a_kc ? 1 : p.sa_k;
    const long long sb_row = b_kc ? p.sb_n : 1, sb_kk = b_kc ? 1 : p.sb_k;
    f32x4 acc[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    constexpr int NS = kBK / 16;  // 16-deep sub-chunks per thread: NS 16-byte loads per operand and chunk
    int ra, ka, rb, kb;
    const int nchunk = (p.K + kBK - 1) / kBK;
    auto load = [&](float4* va, float4* vb, unsigned& okm, int c) {
        c = min(c, nchunk - 1);   // past the end: re-read the last chunk (never staged) rather than branch
        okm = 0u;
        #pragma unroll
        for (int u = 0; u < NS; ++u) {
            if constexpr (VEC) {
                bool oa, ob;
                va[u] = gemm_fetch_vec(p.A, sa_row, sa_kk, m0, p.M, c * kBK + 16 * u, p.K, tid, ra, ka, oa);
                vb[u] = gemm_fetch_vec(p.B, sb_row, sb_kk, n0, p.N, c * kBK + 16 * u, p.K, tid, rb, kb, ob);
                okm |= (oa ? 1u : 0u) << (2 * u) | (ob ? 2u : 0u) << (2 * u);
            } else {
                va[u] = gemm_fetch(p.A, sa_row, sa_kk, m0, p.M, c * kBK + 16 * u, p.K, tid, ra, ka);
                vb[u] = gemm_fetch(p.B, sb_row, sb_kk, n0, p.N, c * kBK + 16 * u, p.K, tid, rb, kb);
            }
        }
    };
    auto stage = [&](int buf, const float4* va, const float4* vb, unsigned okm) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        #pragma unroll
        for (int u = 0; u < NS; ++u) {
            gemm_stage(sA[buf] + 16 * u * kLd, (!VEC || ((okm >> (2 * u)) & 1u)) ? va[u] : z, a_kc, ra, ka);
            gemm_stage(sB[buf] + 16 * u * kLd, (!VEC || ((okm >> (2 * u)) & 2u)) ? vb[u] : z, b_kc, rb, kb);
        }
    };
    auto mma = [&](int buf) {
        #pragma unroll
        for (int ks = 0; ks < kBK; ks += 4) {
            const float av = sA[buf][(ks + kq) * kLd + wave * 16 + r];
            asum += av;
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bv = sB[buf][(ks + kq) * kLd + j * 16 + r];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
            }
        }
    };
    float4 va0[NS], vb0[NS], va1[NS], vb1[NS];
    unsigned ok0, ok1;
    load(va0, vb0, ok0, 0);
    load(va1, vb1, ok1, 1);
    stage(0, va0, vb0, ok0);
    __syncthreads();
    for (int c = 0; c < nchunk; c += 2) {
        load(va0, vb0, ok0, c + 2);
        mma(0);
        if (c + 1 >= nchunk) break;
        stage(1, va1, vb1, ok1);
        __syncthreads();
        load(va1, vb1, ok1, c + 3);
        mma(1);
        if (c + 2 >= nchunk) break;
        stage(0, va0, vb0, ok0);
        __syncthreads();
    }
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 16 * j + r;
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + wave * 16 + kq * 4 + q;
            if (m < p.M) p.C[(long long)m * p.sc_m + (long long)n * p.sc_n] = p.alpha * (acc[j][q] + bias);
        }
    }
    if (p.rowsum && n0 == 0) {  // lanes r, r+16, r+32, r+48 hold the four k-quarters of row m0+16*wave+r
        asum += __shfl_xor(asum, 16);
        asum += __shfl_xor(asum, 32);
        const int m = m0 + wave * 16 + r;
        if (kq == 0 && m < p.M) p.rowsum[m] = p.alpha * asum;
    }
}

__global__ __launch_bounds__(256) void gemm_batch_kernel(GemmArgs a) {
    __shared__ float sA[2][kBK * kLd];
    __shared__ float sB[2][kBK * kLd];
    const int tile = blockIdx.x;
    int pi = 0;
    #pragma unroll
    for (int i = 1; i < kMaxProb; ++i) pi += (i < a.np && tile >= a.p[i].tile0) ? 1 : 0;
    const GemmProb& p = a.p[pi];
    const int t = tile - p.tile0;
    const int m0 = (t / p.tiles_n) * kBM, n0 = (t % p.tiles_n) * kBN;
    if (p.vec) gemm_tile<true>(p, m0, n0, sA, sB);
    else gemm_tile<false>(p, m0, n0, sA, sB);
}

// ------------------------------------------------------------------------------------------- attention core
constexpr int kD = 32;        // head dim (E / heads = 256 / 8)
constexpr int kTile = 64;     // rows per tile = lanes
constexpr int kRow = kD + 4;  // padded LDS row (floats): 16-byte aligned, breaks the 128-byte bank stride

struct AttnArgs {
    const float* Q; const float* K; const float* V;   // (Lq,T,E) pre-scaled q, (Lk,T,E), (Lk,T,E)
    float* O; float* lse;                             // (L,T,E), (L,T,H)
    const float* dO;                                  // bwd
    float* dQ; float* dK; float* dV;                  // bwd: (Lq,T,E), (Lk,T,E), (Lk,T,E)
    const int32_t* img_off;
    int Lq, Lk, L, T, E, H, B;
};

__device__ __forceinline__ void load_row(float* dst, const float* src) {
    #pragma unroll
    for (int c = 0; c < kD; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(src + c);
        dst[c] = v.x; dst[c + 1] = v.y; dst[c + 2] = v.z; dst[c + 3] = v.w;
    }
}
__device__ __forceinline__ void store_row(float* dst, const float* src) {
    #pragma unroll
    for (int c = 0; c < kD; c += 4) *reinterpret_cast<float4*>(dst + c) = make_float4(src[c], src[c + 1], src[c + 2], src[c + 3]);
}
__device__ __forceinline__ float dot_lds(const float* reg, const float* lds_row) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains
    #pragma unroll
    for (int c = 0; c < kD; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(lds_row + c);  // same address in every lane: LDS broadcast
        s0 = fmaf(reg[c], v.x, s0); s1 = fmaf(reg[c + 1], v.y, s1); s2 = fmaf(reg[c + 2], v.z, s2); s3 = fmaf(reg[c + 3], v.w, s3);
    }
    return (s0 + s1) + (s2 + s3);
}

// forward: one wave per (level, image, head); lanes = queries; K/V tiles broadcast from LDS; online softmax per lane
__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float sK[kTile * kRow];
    __shared__ __attribute__((aligned(16))) float sV[kTile * kRow];
    const int h = blockIdx.x % a.H, b = (blockIdx.x / a.H) % a.B, l = blockIdx.x / (a.H * a.B), lane = threadIdx.x;
    const int t0 = a.img_off[b], n = a.img_off[b + 1] - t0;
    const size_t lvl = (size_t)a.T * a.E;
    const float* Q = a.Q + (a.Lq == 1 ? 0 : l) * lvl + (size_t)t0 * a.E + h * kD;
    const float* K = a.K + (a.Lk == 1 ? 0 : l) * lvl + (size_t)t0 * a.E + h * kD;
    const float* V = a.V + (a.Lk == 1 ? 0 : l) * lvl + (size_t)t0 * a.E + h * kD;
    for (int q0 = 0; q0 < n; q0 += kTile) {
        const int i = q0 + lane;
        const bool qok = i < n;
        float q[kD], o[kD];
        if (qok) load_row(q, Q + (size_t)i * a.E);
        #pragma unroll
        for (int c = 0; c < kD; ++c) { o[c] = 0.f; if (!qok) q[c] = 0.f; }
        float m = -INFINITY, s = 0.f;
        for (int k0 = 0; k0 < n; k0 += kTile) {
            const int nk = min(kTile, n - k0);
            __syncthreads();
            if (lane < nk) {
                float tmp[kD];
                load_row(tmp, K + (size_t)(k0 + lane) * a.E); store_row(sK + lane * kRow, tmp);
                load_row(tmp, V + (size_t)(k0 + lane) * a.E); store_row(sV + lane * kRow, tmp);
            }
            __syncthreads();
            for (int j = 0; j < nk; ++j) {
                const float sc = dot_lds(q, sK + j * kRow);
                const float mn = fmaxf(m, sc);
                const float corr = expf(m - mn), pj = expf(sc - mn);
                s = s * corr + pj;
                #pragma unroll
                for (int c = 0; c < kD; c += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(sV + j * kRow + c);
                    o[c] = fmaf(pj, v.x, o[c] * corr); o[c + 1] = fmaf(pj, v.y, o[c + 1] * corr);
                    o[c + 2] = fmaf(pj, v.z, o[c + 2] * corr); o[c + 3] = fmaf(pj, v.w, o[c + 3] * corr);
                }
                m = mn;
            }
        }
        if (qok) {
            const float inv = 1.f / s;
            #pragma unroll
            for (int c = 0; c < kD; ++c) o[c] *= inv;
            store_row(a.O + l * lvl + (size_t)(t0 + i) * a.E + h * kD, o);
            a.lse[((size_t)l * a.T + t0 + i) * a.H + h] = m + logf(s);
        }
    }
}

// backward: one wave per (level, image, head); phase A lanes = queries (dQ), phase B lanes = keys (dK, dV);
// p_ij = exp(s_ij - lse_i).  Outputs are PER-LEVEL partials (L,T,E): the caller sums over levels for an operand
// that was shared by all levels (deterministic, no atomics).
__global__ __launch_bounds__(64) void attn_bwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[kTile * kRow];   // phase A: K rows | phase B: Q rows
    __shared__ __attribute__((aligned(16))) float sB[kTile * kRow];   // phase A: V rows | phase B: dO rows
    __shared__ float sLse[kTile], sDelta[kTile];
    const int h = blockIdx.x % a.H, b = (blockIdx.x / a.H) % a.B, l = blockIdx.x / (a.H * a.B), lane = threadIdx.x;
    const int t0 = a.img_off[b], n = a.img_off[b + 1] - t0;
    const size_t lvl = (size_t)a.T * a.E;
    const size_t base = (size_t)t0 * a.E + h * kD;
    const int lq = a.Lq == 1 ? 0 : l, lk = a.Lk == 1 ? 0 : l;

    // ---- phase A: dQ_i = sum_j dS_ij K_j
    for (int q0 = 0; q0 < n; q0 += kTile) {
        const int i = q0 + lane;
        const bool qok = i < n;
        float dq[kD], q[kD], go[kD];
        float lse = 0.f, delta = 0.f;
        #pragma unroll
        for (int c = 0; c < kD; ++c) dq[c] = 0.f;
        if (qok) {
            load_row(q, a.Q + lq * lvl + base + (size_t)i * a.E);
            load_row(go, a.dO + l * lvl + base + (size_t)i * a.E);
            float o[kD];
            load_row(o, a.O + l * lvl + base + (size_t)i * a.E);
            #pragma unroll
            for (int c = 0; c < kD; ++c) delta = fmaf(go[c], o[c], delta);
            lse = a.lse[((size_t)l * a.T + t0 + i) * a.H + h];
        } else {
            #pragma unroll
            for (int c = 0; c < kD; ++c) { q[c] = 0.f; go[c] = 0.f; }
        }
        for (int k0 = 0; k0 < n; k0 += kTile) {
            const int nk = min(kTile, n - k0);
            __syncthreads();
            if (lane < nk) {
                float tmp[kD];
                load_row(tmp, a.K + lk * lvl + base + (size_t)(k0 + lane) * a.E); store_row(sA + lane * kRow, tmp);
                load_row(tmp, a.V + lk * lvl + base + (size_t)(k0 + lane) * a.E); store_row(sB + lane * kRow, tmp);
            }
            __syncthreads();
            for (int j = 0; j < nk; ++j) {
                const float p = qok ? expf(dot_lds(q, sA + j * kRow) - lse) : 0.f;
                const float ds = p * (dot_lds(go, sB + j * kRow) - delta);
                #pragma unroll
                for (int c = 0; c < kD; c += 4) {
                    const float4 kv = *reinterpret_cast<const float4*>(sA + j * kRow + c);
                    dq[c] = fmaf(ds, kv.x, dq[c]); dq[c + 1] = fmaf(ds, kv.y, dq[c + 1]);
                    dq[c + 2] = fmaf(ds, kv.z, dq[c + 2]); dq[c + 3] = fmaf(ds, kv.w, dq[c + 3]);
                }
            }
        }
        if (qok) store_row(a.dQ + l * lvl + base + (size_t)i * a.E, dq);
    }

    // ---- phase B: dK_j = sum_i dS_ij Q_i, dV_j = sum_i p_ij dO_i
    for (int k0 = 0; k0 < n; k0 += kTile) {
        const int j = k0 + lane;
        const bool kok = j < n;
        float dk[kD], dv[kD], kr[kD], vr[kD];
        #pragma unroll
        for (int c = 0; c < kD; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        if (kok) { load_row(kr, a.K + lk * lvl + base + (size_t)j * a.E); load_row(vr, a.V + lk * lvl + base + (size_t)j * a.E); }
        else {
            #pragma unroll
            for (int c = 0; c < kD; ++c) { kr[c] = 0.f; vr[c] = 0.f; }
        }
        for (int q0 = 0; q0 < n; q0 += kTile) {
            const int nq = min(kTile, n - q0);
            __syncthreads();
            if (lane < nq) {
                float tq[kD], tg[kD], to[kD];
                load_row(tq, a.Q + lq * lvl + base + (size_t)(q0 + lane) * a.E);
                load_row(tg, a.dO + l * lvl + base + (size_t)(q0 + lane) * a.E);
                load_row(to, a.O + l * lvl + base + (size_t)(q0 + lane) * a.E);
                float delta = 0.f;
                #pragma unroll
                for (int c = 0; c < kD; ++c) delta = fmaf(tg[c], to[c], delta);
                store_row(sA + lane * kRow, tq); store_row(sB + lane * kRow, tg);
                sDelta[lane] = delta;
                sLse[lane] = a.lse[((size_t)l * a.T + t0 + q0 + lane) * a.H + h];
            }
            __syncthreads();
            for (int i = 0; i < nq; ++i) {
                const float p = kok ? expf(dot_lds(kr, sA + i * kRow) - sLse[i]) : 0.f;
                const float ds = p * (dot_lds(vr, sB + i * kRow) - sDelta[i]);
                #pragma unroll
                for (int c = 0; c < kD; c += 4) {
                    const float4 qv = *reinterpret_cast<const float4*>(sA + i * kRow + c);
                    const float4 gv = *reinterpret_cast<const float4*>(sB + i * kRow + c);
                    dk[c] = fmaf(ds, qv.x, dk[c]); dk[c + 1] = fmaf(ds, qv.y, dk[c + 1]);
                    dk[c + 2] = fmaf(ds, qv.z, dk[c + 2]); dk[c + 3] = fmaf(ds, qv.w, dk[c + 3]);
                    dv[c] = fmaf(p, gv.x, dv[c]); dv[c + 1] = fmaf(p, gv.y, dv[c + 1]);
                    dv[c + 2] = fmaf(p, gv.z, dv[c + 2]); dv[c + 3] = fmaf(p, gv.w, dv[c + 3]);
                }
            }
        }
        if (kok) { store_row(a.dK + l * lvl + base + (size_t)j * a.E, dk); store_row(a.dV + l * lvl + base + (size_t)j * a.E, dv); }
    }
}

}  // namespace lgd

extern "C" {

// problems: np x 16 doubles? -- plain C struct array, see include/lgd_hip.h (lgd_gemm_problem)
int lgd_gemm_batch(const lgd_gemm_problem* probs_host, int np, void* stream) {
    if (!probs_host || np < 1 || np > lgd::kMaxProb) return LGD_EINVAL;
    lgd::GemmArgs a;
    a.np = np;
    int tile = 0;
    for (int i = 0; i < np; ++i) {
        const lgd_gemm_problem& s = probs_host[i];
        if (!s.A || !s.B || !s.C || s.M < 0 || s.N < 1 || s.K < 1) return LGD_EINVAL;
        lgd::GemmProb& p = a.p[i];
        p.A = s.A; p.B = s.B; p.bias = s.bias; p.C = s.C; p.rowsum = s.rowsum;
        p.M = s.M; p.N = s.N; p.K = s.K;
        p.sa_m = s.sa_m; p.sa_k = s.sa_k; p.sb_n = s.sb_n; p.sb_k = s.sb_k; p.sc_m = s.sc_m; p.sc_n = s.sc_n;
        p.alpha = s.alpha;
        if (!((s.sa_m == 1 || s.sa_k == 1) && (s.sb_n == 1 || s.sb_k == 1))) return LGD_EINVAL;  // one contiguous axis each
        auto vec_ok = [](const float* base, long long s_row, long long s_k, int nrows, int K) {
            if ((reinterpret_cast<size_t>(base) & 15) != 0) return false;
            return s_k == 1 ? (K % 4 == 0 && K >= 4 && s_row % 4 == 0) : (nrows % 4 == 0 && nrows >= 4 && s_k % 4 == 0);
        };
        p.vec = (vec_ok(s.A, s.sa_m, s.sa_k, s.M, s.K) && vec_ok(s.B, s.sb_n, s.sb_k, s.N, s.K)) ? 1 : 0;
        p.tile0 = tile; p.tiles_n = (s.N + lgd::kBN - 1) / lgd::kBN;
        tile += ((s.M + lgd::kBM - 1) / lgd::kBM) * p.tiles_n;
    }
    for (int i = np; i < lgd::kMaxProb; ++i) { a.p[i] = a.p[0]; a.p[i].tile0 = 0x7fffffff; }
    a.ntiles = tile;
    if (tile == 0) return LGD_OK;
    LGD_LAUNCH("gemm_batch_kernel", lgd::gemm_batch_kernel, dim3(tile), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

static int attn_fill(lgd::AttnArgs& a, const float* q, const float* k, const float* v, const int32_t* img_off, int Lq, int Lk,
                     int B, int T, int E, int H) {
    if (!q || !k || !v || !img_off || B < 1 || T < 1 || H < 1 || E != H * lgd::kD || Lq < 1 || Lk < 1 ||
        !(Lq == Lk || Lq == 1 || Lk == 1))
        return LGD_EINVAL;
    a.Q = q; a.K = k; a.V = v; a.img_off = img_off;
    a.Lq = Lq; a.Lk = Lk; a.L = Lq > Lk ? Lq : Lk; a.T = T; a.E = E; a.H = H; a.B = B;
    a.O = nullptr; a.lse = nullptr; a.dO = nullptr; a.dQ = a.dK = a.dV = nullptr;
    return LGD_OK;
}

int lgd_attn_fwd(const float* q, const float* k, const float* v, const int32_t* img_off, int Lq, int Lk, int B, int T, int E,
                 int H, float* out, float* lse, void* stream) {
    lgd::AttnArgs a;
    if (attn_fill(a, q, k, v, img_off, Lq, Lk, B, T, E, H) != LGD_OK || !out || !lse) return LGD_EINVAL;
    a.O = out; a.lse = lse;
    LGD_LAUNCH("attn_fwd_kernel", lgd::attn_fwd_kernel, dim3(B * H * a.L), dim3(64), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

int lgd_attn_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* dout,
                 const int32_t* img_off, int Lq, int Lk, int B, int T, int E, int H, float* dq, float* dk, float* dv,
                 void* stream) {
    lgd::AttnArgs a;
    if (attn_fill(a, q, k, v, img_off, Lq, Lk, B, T, E, H) != LGD_OK || !out || !lse || !dout || !dq || !dk || !dv)
        return LGD_EINVAL;
    a.O = const_cast<float*>(out); a.lse = const_cast<float*>(lse); a.dO = dout; a.dQ = dq; a.dK = dk; a.dV = dv;
    LGD_LAUNCH("attn_bwd_kernel", lgd::attn_bwd_kernel, dim3(B * H * a.L), dim3(64), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
