// Go / no-go prototype of a FUSED Winograd-domain convolution, forward, F(4x4,3x3), fp32, the dominant shape of the step
// (256 -> 256 channels over p3: N=8, 100x168 -> 8400 tiles): input transform on operand load, one v_mfma_f32_16x16x4_f32 chain per
// frequency, output transform from the accumulator registers -- V and M never touch HBM.  (the product runs
// transform kernel -> library GEMM -> transform kernel.)
//   workgroup = 64 output channels x 16 tiles x 36 frequencies, 4 waves, wave w owns co rows [16 w, +16): 36 MFMA tiles x 4 = 144
//     accumulator registers per lane, 1 wave per SIMD.  Measured alternatives: 64 co x 32 tiles on 4 waves = 288 accumulators per lane --
//     more than the 256 AccVGPRs, hipcc parks them in VGPRs and copies them through a few AccVGPRs around EVERY MFMA: 15.3 TF; the same
//     block on 8 waves (144 each, but only 256 registers per wave at 2 waves per SIMD): 164 spills, 10.9 TF.
//   per K-step of 8 input channels: thread (tile, channel) loads its 6x6 window and transforms it into LDS V[f][k][tile];
//     U[f][k][co] (host layout [Ci/8][36][8][Co]) goes global -> registers -> LDS; then 36 x 2 = 72 MFMAs per wave;
//     the NEXT step's window / U loads are issued before the MFMA phase and consumed after it (register staging).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/fused_wino_lab tools/lab/fused_wino_lab.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CO_BLK = 64, T_BLK = 16, KC = 8, NF = 36;
constexpr int U_LD = 80, V_LD = 16;   // padded LDS rows: the 4 x 16 operand patches of an MFMA read 64 distinct banks

__device__ __forceinline__ void bt6(const float* d, float* t) {
    const float a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = a + b; t[2] = a - b; t[3] = c + e; t[4] = c - e;
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
__device__ __forceinline__ void at6(const float* m, float* y) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34; y[1] = d12 + 2.f * d34; y[2] = s12 + 4.f * s34; y[3] = d12 + 8.f * d34 + m[5];
}

struct Args {
    const float* x; const float* Ut; float* y;
    int N, C, H, W, TH, TW, Co;
    long long T;
};

template <bool PREFETCH, int MODE>   // MODE 0: the convolution; 1: MFMA phases only (no loads, no staging); 2: loads + transforms + staging only
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fused_wino_fwd(Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sU = lds;                      // [NF][KC][U_LD]
    float* sV = lds + NF * KC * U_LD;     // [NF][KC][V_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long t0 = (long long)blockIdx.x * T_BLK;
    const int co0 = blockIdx.y * CO_BLK;
    // transform role (threads 0..255): (tile tt, channel kk of the K-step)
    const bool xf = tid < 128;
    const int tt = tid & 15, kk = (tid >> 4) & 7;
    const long long t = t0 + tt;
    const bool tok = t < a.T;
    const long long tc = tok ? t : a.T - 1;
    const int tx = (int)(tc % a.TW), ty = (int)((tc / a.TW) % a.TH), n = (int)(tc / ((long long)a.TW * a.TH));
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    // clamped window offsets + validity mask (zero padding applied at transform time, not at load time: nothing waits on the loads)
    int roff[6], coff[6];
    unsigned long long okm = 0ull;
    #pragma unroll
    for (int i = 0; i < 6; ++i) {
        roff[i] = min(max(y0 + i, 0), a.H - 1) * a.W;
        coff[i] = min(max(x0 + i, 0), a.W - 1);
    }
    #pragma unroll
    for (int i = 0; i < 6; ++i)
        #pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int yy = y0 + i, xx = x0 + j;
            const bool ok = tok && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            okm |= (ok ? 1ull : 0ull) << (6 * i + j);
        }
    const size_t plane = (size_t)a.H * a.W;
    const float* xin = a.x + (size_t)n * a.C * plane;
    // U role: 36 * 8 rows of 64 floats = 4608 float4, 18 per thread: row = idx / 16, quad = idx % 16
    f32x4 acc[NF];
    #pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = {0.f, 0.f, 0.f, 0.f};
    float win[36];
    float4 ureg[18];
    auto issue = [&](int c0) {
        const float* p = xin + (size_t)(c0 + kk) * plane;
        if (xf) {
            #pragma unroll
            for (int i = 0; i < 6; ++i)
                #pragma unroll
                for (int j = 0; j < 6; ++j) win[6 * i + j] = p[roff[i] + coff[j]];
        }
        const float* up = a.Ut + (size_t)(c0 / KC) * NF * KC * a.Co + co0;
        #pragma unroll
        for (int q = 0; q < 18; ++q) {
            const int idx = q * 256 + tid, row = idx >> 4, quad = idx & 15;
            ureg[q] = *reinterpret_cast<const float4*>(up + (size_t)row * a.Co + quad * 4);
        }
    };
    auto stage = [&]() {
        #pragma unroll
        for (int q = 0; q < 18; ++q) {
            const int idx = q * 256 + tid, row = idx >> 4, quad = idx & 15;
            *reinterpret_cast<float4*>(&sU[row * U_LD + quad * 4]) = ureg[q];
        }
        if (!xf) return;
        float d[6][6], r[6][6];
        #pragma unroll
        for (int i = 0; i < 6; ++i)
            #pragma unroll
            for (int j = 0; j < 6; ++j) d[i][j] = ((okm >> (6 * i + j)) & 1ull) ? win[6 * i + j] : 0.f;
        #pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
            float w[6];
            bt6(col, w);
            #pragma unroll
            for (int i = 0; i < 6; ++i) r[i][j] = w[i];
        }
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            float w[6];
            bt6(r[i], w);
            #pragma unroll
            for (int j = 0; j < 6; ++j) sV[((6 * i + j) * KC + kk) * V_LD + tt] = w[j];
        }
    };
    const int r16 = lane & 15, kq = lane >> 4, cq = wave & 3, th = 0;
    auto mma = [&]() {
        // operands of 6 frequencies (12 MFMAs) are read from LDS as one batch, then their MFMAs issue back to back: the asm statements
        // pin the accumulators but also pin the order, so the LDS latency is paid once per batch instead of once per MFMA
        #pragma unroll
        for (int fg = 0; fg < NF; fg += 6) {
            float av[6][KC / 4], bv[6][KC / 4];
            #pragma unroll
            for (int i = 0; i < 6; ++i)
                #pragma unroll
                for (int k4 = 0; k4 < KC / 4; ++k4) {
                    const int row = (fg + i) * KC + k4 * 4 + kq;
                    av[i][k4] = sU[row * U_LD + cq * 16 + r16];
                    bv[i][k4] = sV[row * V_LD + th * 16 + r16];
                }
            #pragma unroll
            for (int i = 0; i < 6; ++i)
                #pragma unroll
                for (int k4 = 0; k4 < KC / 4; ++k4)
                    // accumulators pinned in AccVGPRs ("+a"): with the builtin hipcc keeps them in VGPRs and copies each through an
                    // AccVGPR around every MFMA (183 reads + 315 writes per K-step, 10.7 TF)
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[fg + i]) : "v"(av[i][k4]), "v"(bv[i][k4]));
        }
    };
    issue(0);
    for (int c0 = 0; c0 < a.C; c0 += KC) {
        if (c0) __syncthreads();          // the previous MFMA phase is done with the slabs
        if (MODE != 1) stage();
        __syncthreads();
        if (MODE != 1) { if (PREFETCH) { if (c0 + KC < a.C) issue(c0 + KC); } }
        if (MODE != 2) mma();
        if (MODE != 1) { if (!PREFETCH) { if (c0 + KC < a.C) issue(c0 + KC); } }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results must have landed before the compiler reads the AccVGPRs
    // output transform from the accumulators: lane holds tile column j = lane & 15 (+16), channel rows (lane >> 4) * 4 + reg
    {
        const long long to = t0 + th * 16 + r16;
        if (to >= a.T) return;
        const int ox = (int)(to % a.TW) * 4, oy = (int)((to / a.TW) % a.TH) * 4, on = (int)(to / ((long long)a.TW * a.TH));
        #pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int co = co0 + cq * 16 + kq * 4 + reg;
            float rr[4][6];
            #pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float col[6] = {acc[j][reg], acc[6 + j][reg], acc[12 + j][reg], acc[18 + j][reg], acc[24 + j][reg], acc[30 + j][reg]};
                float w[4];
                at6(col, w);
                rr[0][j] = w[0]; rr[1][j] = w[1]; rr[2][j] = w[2]; rr[3][j] = w[3];
            }
            float* yo = a.y + ((size_t)on * a.Co + co) * plane;
            #pragma unroll
            for (int i = 0; i < 4; ++i) {
                float o[4];
                at6(rr[i], o);
                if (oy + i < a.H) *reinterpret_cast<float4*>(yo + (size_t)(oy + i) * a.W + ox) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

static const double Gm[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};

int main() {
    const int N = 8, C = 256, Co = 256, H = 100, W = 168, TH = H / 4, TW = W / 4;
    const long long T = (long long)N * TH * TW;
    std::vector<float> hx((size_t)N * C * H * W), hw((size_t)Co * C * 9), hUt((size_t)(C / KC) * NF * KC * Co);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = 0.05f * rnd();
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < C; ++ci) {
            const float* g = &hw[((size_t)co * C + ci) * 9];
            double tmp[6][3];
            for (int a_ = 0; a_ < 6; ++a_) for (int j = 0; j < 3; ++j) { double v = 0; for (int i = 0; i < 3; ++i) v += Gm[a_][i] * g[3 * i + j]; tmp[a_][j] = v; }
            for (int a_ = 0; a_ < 6; ++a_) for (int b = 0; b < 6; ++b) {
                double v = 0; for (int j = 0; j < 3; ++j) v += tmp[a_][j] * Gm[b][j];
                hUt[(((size_t)(ci / KC) * NF + (6 * a_ + b)) * KC + ci % KC) * Co + co] = (float)v;
            }
        }
    float *dx, *dU, *dy;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dU, hUt.size() * 4)); CK(hipMalloc(&dy, (size_t)N * Co * H * W * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dU, hUt.data(), hUt.size() * 4, hipMemcpyHostToDevice));
    Args a{dx, dU, dy, N, C, H, W, TH, TW, Co, T};
    const size_t smem = (size_t)(NF * KC * (U_LD + V_LD)) * sizeof(float);
    CK(hipFuncSetAttribute((const void*)fused_wino_fwd<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute((const void*)fused_wino_fwd<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute((const void*)fused_wino_fwd<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute((const void*)fused_wino_fwd<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const dim3 grid((unsigned)((T + T_BLK - 1) / T_BLK), Co / CO_BLK), block(256);
    printf("grid %u x %u workgroups (%.2f rounds of 256 CUs), LDS %zu B per workgroup\n", grid.x, grid.y, grid.x * grid.y / 256.0, smem);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int var = 0; var < 4; ++var) {
        auto launch = [&]() {
            if (var == 0) hipLaunchKernelGGL((fused_wino_fwd<false, 0>), grid, block, smem, 0, a);
            else if (var == 1) hipLaunchKernelGGL((fused_wino_fwd<true, 0>), grid, block, smem, 0, a);
            else if (var == 2) hipLaunchKernelGGL((fused_wino_fwd<true, 1>), grid, block, smem, 0, a);
            else hipLaunchKernelGGL((fused_wino_fwd<true, 2>), grid, block, smem, 0, a);
        };
        CK(hipMemset(dy, 0, (size_t)N * Co * H * W * 4));
        launch(); launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / reps;
        const double fl = 2.0 * NF * Co * C * (double)T, fl_direct = 2.0 * 9 * Co * C * (double)N * H * W;
        // validation on sampled outputs against the direct convolution in fp64
        std::vector<float> hy((size_t)N * Co * H * W);
        CK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int k = 0; k < 400; ++k) {
            const int n = k % N, co = (k * 37) % Co, yy = (k * 13) % H, xx = (k * 29) % W;
            double ref = 0;
            for (int ci = 0; ci < C; ++ci)
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) {
                        const int y2 = yy + i - 1, x2 = xx + j - 1;
                        if (y2 < 0 || y2 >= H || x2 < 0 || x2 >= W) continue;
                        ref += (double)hw[((size_t)co * C + ci) * 9 + 3 * i + j] * hx[(((size_t)n * C + ci) * H + y2) * W + x2];
                    }
            maxerr = fmax(maxerr, fabs(ref - hy[(((size_t)n * Co + co) * H + yy) * W + xx]));
            maxref = fmax(maxref, fabs(ref));
        }
        printf("%-22s %8.1f us   %6.1f TFLOP/s on the frequency-domain products (%.0f TF direct-conv equivalent)   max err %.2e of %.2f\n",
               var == 0 ? "fused, no prefetch" : var == 1 ? "fused, reg-staged" : var == 2 ? "  MFMA phases only" : "  load+transform only", us, fl / us / 1e6, fl_direct / us / 1e6, maxerr, maxref);
    }
    printf("product pipeline at this shape (profiles/r02_kbench_hbm_cold.log scaled to 8400 tiles): in ~95 + GEMM ~317 (125 TF) + out ~78 = ~490 us\n");
    return 0;
}
