// F(4x4,3x3) input-transform store-shape lab (p3: N=8, C=256, 100x168 -> 8400 tiles, 36 planes)
//  A: thread per tile, 36 scalar NT stores (256 B per wave and plane)            [product shape]
//  B: block's 256 tiles staged through LDS, each plane written as 1 KB (float4 per lane)
//  C: two tiles per thread, float2 NT stores (512 B per wave and plane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bt6(const float* d, float* t) {
    const float a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = a + b; t[2] = a - b; t[3] = c + e; t[4] = c - e;
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

__device__ __forceinline__ void load_window(const float* p, int H, int W, int TW, int tx, int ty, float (&d)[6][6]) {
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1, lane = threadIdx.x & 63;
    #pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int y = y0 + i;
        const bool yok = y >= 0 && y < H;
        const float* row = p + (size_t)(yok ? y : 0) * W;
        float4 m = yok ? *reinterpret_cast<const float4*>(row + x0 + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
        float e0 = __shfl_up(m.w, 1), e5 = __shfl_down(m.x, 1);
        if (lane == 0 && tx != 0) e0 = yok ? row[x0] : 0.f;
        if (lane == 63 && tx != TW - 1) e5 = yok ? row[x0 + 5] : 0.f;
        if (tx == 0) e0 = 0.f;
        if (tx == TW - 1) e5 = 0.f;
        d[i][0] = e0; d[i][1] = m.x; d[i][2] = m.y; d[i][3] = m.z; d[i][4] = m.w; d[i][5] = e5;
    }
}

template <int VAR, int LAYOUT = 0, int ITER = 1>
__global__ __launch_bounds__(256) void wino4_in(const float* __restrict__ x, float* __restrict__ V, int N, int C, int H, int W) {
    const int TH = H / 4, TW = W / 4;
    const long long T = (long long)N * TH * TW;
    const int c = blockIdx.y;
    const size_t plane = LAYOUT ? (size_t)T : (size_t)C * T;
    const size_t cstride = LAYOUT ? (size_t)36 * T : (size_t)T;
    __shared__ float lds[VAR == 1 ? 36 * 256 : 1];
    if (VAR == 2) {
        const int TWP = TW / 2;
        const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
        if (u >= (long long)N * TH * TWP) return;
        const int txp = (int)(u % TWP), ty = (int)((u / TWP) % TH), n = (int)(u / ((long long)TWP * TH));
        const float* p = x + ((size_t)n * C + c) * H * W;
        const int y0 = 4 * ty - 1, x0 = 8 * txp - 1, lane = threadIdx.x & 63;
        float d[6][10];
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int y = y0 + i;
            const bool yok = y >= 0 && y < H;
            const float* row = p + (size_t)(yok ? y : 0) * W;
            const float4 a = yok ? *reinterpret_cast<const float4*>(row + x0 + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 b = yok ? *reinterpret_cast<const float4*>(row + x0 + 5) : make_float4(0.f, 0.f, 0.f, 0.f);
            float e0 = __shfl_up(b.w, 1), e9 = __shfl_down(a.x, 1);
            if (lane == 0 && txp != 0) e0 = yok ? row[x0] : 0.f;
            if (lane == 63 && txp != TWP - 1) e9 = yok ? row[x0 + 9] : 0.f;
            if (txp == 0) e0 = 0.f;
            if (txp == TWP - 1) e9 = 0.f;
            d[i][0] = e0; d[i][1] = a.x; d[i][2] = a.y; d[i][3] = a.z; d[i][4] = a.w; d[i][5] = b.x; d[i][6] = b.y; d[i][7] = b.z; d[i][8] = b.w; d[i][9] = e9;
        }
        float* o = V + (size_t)c * cstride + ((size_t)n * TH + ty) * TW + 2 * txp;
        float r[2][6][6];
        #pragma unroll
        for (int q = 0; q < 2; ++q)
            #pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float col[6] = {d[0][4 * q + j], d[1][4 * q + j], d[2][4 * q + j], d[3][4 * q + j], d[4][4 * q + j], d[5][4 * q + j]};
                float w[6]; bt6(col, w);
                #pragma unroll
                for (int i = 0; i < 6; ++i) r[q][i][j] = w[i];
            }
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            float w0[6], w1[6];
            bt6(r[0][i], w0); bt6(r[1][i], w1);
            #pragma unroll
            for (int j = 0; j < 6; ++j) { vf2 v; v.x = w0[j]; v.y = w1[j]; __builtin_nontemporal_store(v, reinterpret_cast<vf2*>(o + (size_t)(6 * i + j) * plane)); }
        }
        return;
    }
  for (int it = 0; it < ITER; ++it) {
    const long long u = ((long long)blockIdx.x * ITER + it) * 256 + threadIdx.x;
    const bool on = u < T;
    const long long uu = on ? u : T - 1;
    const int tx = (int)(uu % TW), ty = (int)((uu / TW) % TH), n = (int)(uu / ((long long)TW * TH));
    const float* p = x + ((size_t)n * C + c) * H * W;
    float d[6][6];
    load_window(p, H, W, TW, tx, ty, d);
    float r[6][6];
    #pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float w[6]; bt6(col, w);
        #pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = w[i];
    }
    if (VAR == 0) {
        if (!on) continue;
        float* o = V + (size_t)c * cstride + u;
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            float w[6]; bt6(r[i], w);
            #pragma unroll
            for (int j = 0; j < 6; ++j) __builtin_nontemporal_store(w[j], o + (size_t)(6 * i + j) * plane);
        }
    } else {
        #pragma unroll
        for (int i = 0; i < 6; ++i) {
            float w[6]; bt6(r[i], w);
            #pragma unroll
            for (int j = 0; j < 6; ++j) lds[(6 * i + j) * 256 + threadIdx.x] = w[j];
        }
        __syncthreads();
        // 36 planes x 256 floats: wave w writes planes w, w+4, ...; lane writes float4 #lane of the plane's 1 KB
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const long long t0 = ((long long)blockIdx.x * ITER + it) * 256;
        #pragma unroll
        for (int f = wave; f < 36; f += 4) {
            const float4 v = *reinterpret_cast<const float4*>(&lds[f * 256 + lane * 4]);
            if (t0 + lane * 4 + 3 < T) {
                vf4 q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
                __builtin_nontemporal_store(q, reinterpret_cast<vf4*>(V + (size_t)f * plane + (size_t)c * T + t0 + lane * 4));
            }
        }
    }
    if (VAR == 1) __syncthreads();
  }
}

// staged variants on the [c][f][t] layout: NT threads per block, PH phases of 36/PH planes each (LDS = 36/PH * NT * 4 B)
template <int NT, int PH, int G = 0, bool SWAP = false, bool PLAIN = false>
__global__ __launch_bounds__(NT) void wino4_in_staged(const float* __restrict__ x, float* __restrict__ V, int N, int C, int H, int W, long long TS = 0) {
    const int TH = H / 4, TW = W / 4;
    const long long T = (long long)N * TH * TW;
    const int c = SWAP ? blockIdx.x : blockIdx.y;
    constexpr int NPL = 36 / PH;
    __shared__ __attribute__((aligned(16))) float lds[NPL * NT];
    // G > 0: workgroups are dealt round-robin to the 8 XCDs; give each XCD runs of G consecutive tile blocks
    long long lb = SWAP ? blockIdx.y : blockIdx.x;
    if (G > 0) { const long long xcd = lb % 8, idx = lb / 8; lb = ((idx / G) * 8 + xcd) * G + idx % G; }
    const long long t0 = lb * NT;
    const long long u = t0 + threadIdx.x;
    const bool on = u < T;
    const long long uu = on ? u : T - 1;
    const int tx = (int)(uu % TW), ty = (int)((uu / TW) % TH), n = (int)(uu / ((long long)TW * TH));
    const float* p = x + ((size_t)n * C + c) * H * W;
    float d[6][6];
    load_window(p, H, W, TW, tx, ty, d);
    float r[6][6];
    #pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float w[6]; bt6(col, w);
        #pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = w[i];
    }
    if (TS == 0) TS = T;  // plane stride in floats (TS > T: runs aligned to 128 B / 1 KB)
    float* dst = V + (size_t)c * 36 * TS + t0;
    constexpr int RUN4 = NT / 4;  // float4 per plane run
    #pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
        if (ph) __syncthreads();
        #pragma unroll
        for (int ii = 0; ii < 6 / PH; ++ii) {
            const int i = ph * (6 / PH) + ii;
            float w[6]; bt6(r[i], w);
            #pragma unroll
            for (int j = 0; j < 6; ++j) lds[(6 * ii + j) * NT + threadIdx.x] = w[j];
        }
        __syncthreads();
        // NPL planes x RUN4 float4: thread k handles float4 index k, k+NT, ...
        #pragma unroll
        for (int k = 0; k < NPL * RUN4 / NT; ++k) {
            const int idx = k * NT + threadIdx.x;
            const int f = idx / RUN4, q4 = idx % RUN4;
            const float4 v = *reinterpret_cast<const float4*>(&lds[f * NT + q4 * 4]);
            if (t0 + q4 * 4 + 3 < T) {
                vf4 q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
                if (PLAIN) *reinterpret_cast<vf4*>(dst + (size_t)(ph * NPL + f) * TS + q4 * 4) = q;
                else __builtin_nontemporal_store(q, reinterpret_cast<vf4*>(dst + (size_t)(ph * NPL + f) * TS + q4 * 4));
            }
        }
    }
}

int main() {
    const int N = 8, C = 256, H = 100, W = 168;
    const size_t nx = (size_t)N * C * H * W, T = (size_t)N * (H / 4) * (W / 4), nv = 36 * (size_t)C * 16640;   // room for the plane-stride experiments (TS up to 16640 floats)
    const int NB = 3;
    std::vector<float*> X(NB), Vb(NB);
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&X[i], nx * 4)); CK(hipMemset(X[i], 1, nx * 4)); CK(hipMalloc(&Vb[i], nv * 4)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)(nx + nv) * 4;
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch(X[i % NB], Vb[i % NB]);
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e0)); launch(X[i % NB], Vb[i % NB]); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms); tot += ms;
        }
        printf("%-34s avg %7.1f us  best %7.1f us  %.2f TB/s (alg)\n", name, tot / 12 * 1e3, best * 1e3, bytes / (tot / 12 * 1e-3) / 1e12);
    };
    run("A scalar NT stores", [&](float* x, float* v) { wino4_in<0><<<dim3((unsigned)((T + 255) / 256), C), 256>>>(x, v, N, C, H, W); });
    run("B LDS-staged 1 KB stores", [&](float* x, float* v) { wino4_in<1><<<dim3((unsigned)((T + 255) / 256), C), 256>>>(x, v, N, C, H, W); });
    run("A layout [c][f][t]", [&](float* x, float* v) { wino4_in<0, 1><<<dim3((unsigned)((T + 255) / 256), C), 256>>>(x, v, N, C, H, W); });
    run("B layout [c][f][t]", [&](float* x, float* v) { wino4_in<1, 1><<<dim3((unsigned)((T + 255) / 256), C), 256>>>(x, v, N, C, H, W); });
#define ST(NT, PH) run("staged [c][f][t] NT=" #NT " PH=" #PH, [&](float* x, float* v) { wino4_in_staged<NT, PH><<<dim3((unsigned)((T + NT - 1) / NT), C), NT>>>(x, v, N, C, H, W); })
    ST(256, 1); ST(256, 3); ST(256, 6); ST(512, 3); ST(512, 6); ST(1024, 3); ST(1024, 6); ST(128, 3); ST(128, 1);
#define STG(NT, PH, G) run("staged NT=" #NT " PH=" #PH " XCD runs G=" #G, [&](float* x, float* v) { const unsigned nb = (unsigned)((T + NT - 1) / NT); wino4_in_staged<NT, PH, G><<<dim3((nb + 8 * G - 1) / (8 * G) * (8 * G), C), NT>>>(x, v, N, C, H, W); })
    run("staged NT=256 PH=1 channel-fastest dispatch", [&](float* x, float* v) { wino4_in_staged<256, 1, 0, true><<<dim3(C, (unsigned)((T + 255) / 256)), 256>>>(x, v, N, C, H, W); });
    run("staged NT=256 PH=3 channel-fastest dispatch", [&](float* x, float* v) { wino4_in_staged<256, 3, 0, true><<<dim3(C, (unsigned)((T + 255) / 256)), 256>>>(x, v, N, C, H, W); });
#define STS(TSV, PL) run("staged NT=256 PH=3 plane stride " #TSV " plain=" #PL, [&](float* x, float* v) { wino4_in_staged<256, 3, 0, false, PL><<<dim3((unsigned)((T + 255) / 256), C), 256>>>(x, v, N, C, H, W, TSV); })
    // run alignment: T = 8400 -> runs 16 B aligned; 8416 -> 128 B; 8448 -> 1 KB; 8512 -> 256 B; 8404 -> 16 B (control)
    STS(8400, false); STS(8404, false); STS(8416, false); STS(8448, false); STS(8512, false); STS(8400, true); STS(8416, true); STS(8448, true);
    STS(8400, false); STS(8416, false);
    // plane stride vs the HBM channel interleave: 34 / 36 / 40 / 48 / 64 / 65 KB planes
    STS(8704, false); STS(9216, false); STS(10240, false); STS(12288, false); STS(16384, false); STS(16640, false); STS(8400, false);
    run("C two tiles/thread float2", [&](float* x, float* v) { wino4_in<2><<<dim3((unsigned)((T / 2 + 255) / 256), C), 256>>>(x, v, N, C, H, W); });
    return 0;
}
