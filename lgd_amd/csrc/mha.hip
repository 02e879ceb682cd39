// K2: inter-object relation adaptation = nn.MultiheadAttention(E=256, heads=8) between appearance and
// label embeddings with a block-diagonal (per-image) mask  [ref: dynamic_teacher.py:76-78, 255-273].
//
// The reference calls the module once per FPN level (5x), each call re-projecting the shared K/V
// operand and launching ~8 tiny kernels over a (T,T) masked score matrix.  Here:
//   * lgd_gemm_batch : fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32 fma chains) for the dense in/out
//     projections of ALL levels in one launch (a list of strided GEMM problems; also serves the
//     backward's dX / dW / dbias products);
//   * mha core       : one wave64 per (image, head); lanes own query rows (forward, dQ) or key rows
//     (dK, dV), the partner rows are broadcast from LDS, softmax is an online max/sum per lane --
//     only the image's own n x n block is ever touched (no (T,T) mask), levels are looped inside so a
//     broadcast operand's gradient is accumulated in registers (deterministic, no atomics).
// Sizes are tiny (T ~ 10^2 tokens): these kernels are latency-bound, not roofline-bound; MFMA is used
// because the projections are GEMM-shaped, not because it pays.
#include "common.h"

namespace lgd {

// ------------------------------------------------------------------------------------------- GEMM list
constexpr int kMaxProb = 6;
struct GemmProb {
    const float* A; const float* B; const float* bias; float* C; float* rowsum;
    int M, N, K;
    long long sa_m, sa_k, sb_n, sb_k, sc_m, sc_n;
    float alpha;
    int tile0, tiles_n;
};
struct GemmArgs { GemmProb p[kMaxProb]; int np, ntiles; };

using f32x4 = __attribute__((ext_vector_type(4))) float;

// C[m,n] = alpha * (sum_k A(m,k) * B(n,k) + bias[n]);  rowsum[m] = sum_k A(m,k) (optional, n-tile 0 only)
// One wave per 16(m) x 64(n) tile: 4 accumulators of v_mfma_f32_16x16x4_f32 (A: lane l holds
// A[i=l&15][k=l>>4]; B: B[k=l>>4][j=l&15]; C/D: col=l&15, row=(l>>4)*4+reg).
__global__ __launch_bounds__(256) void gemm_batch_kernel(GemmArgs a) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const int lane = threadIdx.x & 63;
    int pi = 0;
    #pragma unroll
    for (int i = 1; i < kMaxProb; ++i) pi += (i < a.np && tile >= a.p[i].tile0) ? 1 : 0;
    const GemmProb& p = a.p[pi];
    const int t = tile - p.tile0;
    const int m0 = (t / p.tiles_n) * 16, n0 = (t % p.tiles_n) * 64;
    const int r = lane & 15, kq = lane >> 4;
    const bool mok = m0 + r < p.M;
    const float* pa = p.A + (long long)(m0 + r) * p.sa_m;
    const float* pb[4];
    bool nok[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) { nok[j] = n0 + 16 * j + r < p.N; pb[j] = p.B + (long long)(n0 + 16 * j + r) * p.sb_n; }
    f32x4 acc[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    #pragma unroll 4
    for (int k0 = 0; k0 < p.K; k0 += 4) {
        const int k = k0 + kq;
        const bool kok = k < p.K;
        const float av = (mok && kok) ? pa[(long long)k * p.sa_k] : 0.f;
        asum += av;
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float bv = (nok[j] && kok) ? pb[j][(long long)k * p.sb_k] : 0.f;
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
        }
    }
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 16 * j + r;
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + kq * 4 + q;
            if (m < p.M) p.C[(long long)m * p.sc_m + (long long)n * p.sc_n] = p.alpha * (acc[j][q] + bias);
        }
    }
    if (p.rowsum && n0 == 0) {  // lanes r, r+16, r+32, r+48 hold the four k-quarters of row m0+r
        asum += __shfl_xor(asum, 16);
        asum += __shfl_xor(asum, 32);
        if (kq == 0 && mok) p.rowsum[m0 + r] = p.alpha * asum;
    }
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
    int Lq, Lk, L, T, E, H;
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
    float s = 0.f;
    #pragma unroll
    for (int c = 0; c < kD; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(lds_row + c);  // same address in every lane: LDS broadcast
        s = fmaf(reg[c], v.x, s); s = fmaf(reg[c + 1], v.y, s); s = fmaf(reg[c + 2], v.z, s); s = fmaf(reg[c + 3], v.w, s);
    }
    return s;
}

// forward: lanes = queries; K/V tiles broadcast from LDS; online softmax per lane
__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float sK[kTile * kRow];
    __shared__ __attribute__((aligned(16))) float sV[kTile * kRow];
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H, lane = threadIdx.x;
    const int t0 = a.img_off[b], n = a.img_off[b + 1] - t0;
    const size_t lvl = (size_t)a.T * a.E;
    for (int l = 0; l < a.L; ++l) {
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
}

// backward: phase A lanes = queries (dQ), phase B lanes = keys (dK, dV); p_ij = exp(s_ij - lse_i)
__global__ __launch_bounds__(64) void attn_bwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[kTile * kRow];   // phase A: K rows | phase B: Q rows
    __shared__ __attribute__((aligned(16))) float sB[kTile * kRow];   // phase A: V rows | phase B: dO rows
    __shared__ float sLse[kTile], sDelta[kTile];
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H, lane = threadIdx.x;
    const int t0 = a.img_off[b], n = a.img_off[b + 1] - t0;
    const size_t lvl = (size_t)a.T * a.E;
    const size_t base = (size_t)t0 * a.E + h * kD;

    // ---- phase A: dQ_i = sum_j dS_ij K_j ; summed over levels when the query operand is shared (Lq == 1)
    for (int q0 = 0; q0 < n; q0 += kTile) {
        const int i = q0 + lane;
        const bool qok = i < n;
        float dq[kD];
        #pragma unroll
        for (int c = 0; c < kD; ++c) dq[c] = 0.f;
        for (int l = 0; l < a.L; ++l) {
            const int lq = a.Lq == 1 ? 0 : l, lk = a.Lk == 1 ? 0 : l;
            float q[kD], go[kD];
            float lse = 0.f, delta = 0.f;
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
            if (a.Lq != 1) {
                if (qok) store_row(a.dQ + l * lvl + base + (size_t)i * a.E, dq);
                #pragma unroll
                for (int c = 0; c < kD; ++c) dq[c] = 0.f;
            }
        }
        if (a.Lq == 1 && qok) store_row(a.dQ + base + (size_t)i * a.E, dq);
    }

    // ---- phase B: dK_j = sum_i dS_ij Q_i, dV_j = sum_i p_ij dO_i ; summed over levels when Lk == 1
    for (int k0 = 0; k0 < n; k0 += kTile) {
        const int j = k0 + lane;
        const bool kok = j < n;
        float dk[kD], dv[kD];
        #pragma unroll
        for (int c = 0; c < kD; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        for (int l = 0; l < a.L; ++l) {
            const int lq = a.Lq == 1 ? 0 : l, lk = a.Lk == 1 ? 0 : l;
            float kr[kD], vr[kD];
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
            if (a.Lk != 1) {
                if (kok) { store_row(a.dK + l * lvl + base + (size_t)j * a.E, dk); store_row(a.dV + l * lvl + base + (size_t)j * a.E, dv); }
                #pragma unroll
                for (int c = 0; c < kD; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
            }
        }
        if (a.Lk == 1 && kok) { store_row(a.dK + base + (size_t)j * a.E, dk); store_row(a.dV + base + (size_t)j * a.E, dv); }
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
        p.tile0 = tile; p.tiles_n = (s.N + 63) / 64;
        tile += ((s.M + 15) / 16) * p.tiles_n;
    }
    for (int i = np; i < lgd::kMaxProb; ++i) { a.p[i] = a.p[0]; a.p[i].tile0 = 0x7fffffff; }
    a.ntiles = tile;
    if (tile == 0) return LGD_OK;
    LGD_LAUNCH("gemm_batch_kernel", lgd::gemm_batch_kernel, dim3((tile + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

static int attn_fill(lgd::AttnArgs& a, const float* q, const float* k, const float* v, const int32_t* img_off, int Lq, int Lk,
                     int B, int T, int E, int H) {
    if (!q || !k || !v || !img_off || B < 1 || T < 1 || H < 1 || E != H * lgd::kD || Lq < 1 || Lk < 1 ||
        !(Lq == Lk || Lq == 1 || Lk == 1))
        return LGD_EINVAL;
    a.Q = q; a.K = k; a.V = v; a.img_off = img_off;
    a.Lq = Lq; a.Lk = Lk; a.L = Lq > Lk ? Lq : Lk; a.T = T; a.E = E; a.H = H;
    a.O = nullptr; a.lse = nullptr; a.dO = nullptr; a.dQ = a.dK = a.dV = nullptr;
    return LGD_OK;
}

int lgd_attn_fwd(const float* q, const float* k, const float* v, const int32_t* img_off, int Lq, int Lk, int B, int T, int E,
                 int H, float* out, float* lse, void* stream) {
    lgd::AttnArgs a;
    if (attn_fill(a, q, k, v, img_off, Lq, Lk, B, T, E, H) != LGD_OK || !out || !lse) return LGD_EINVAL;
    a.O = out; a.lse = lse;
    LGD_LAUNCH("attn_fwd_kernel", lgd::attn_fwd_kernel, dim3(B * H), dim3(64), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

int lgd_attn_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* dout,
                 const int32_t* img_off, int Lq, int Lk, int B, int T, int E, int H, float* dq, float* dk, float* dv,
                 void* stream) {
    lgd::AttnArgs a;
    if (attn_fill(a, q, k, v, img_off, Lq, Lk, B, T, E, H) != LGD_OK || !out || !lse || !dout || !dq || !dk || !dv)
        return LGD_EINVAL;
    a.O = const_cast<float*>(out); a.lse = const_cast<float*>(lse); a.dO = dout; a.dQ = dq; a.dK = dk; a.dV = dv;
    LGD_LAUNCH("attn_bwd_kernel", lgd::attn_bwd_kernel, dim3(B * H), dim3(64), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
