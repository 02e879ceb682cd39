// wino_in access-shape lab (p3: N=8, C=256, 100x168): what bounds the input transform?
//  V0 product shape: float4 + 2 halo scalars per row, 4 rows / tile row
//  V1 no halo columns (wrong numerics; measures the cost of the scalar halo loads)
//  V2 no halo columns, only rows 1,2 (read amplification 1.0)
//  V3 halo columns from the neighbour lanes (ds_bpermute), 4 rows
//  V4 V3 + R tile rows per thread with a rolling 2-row overlap
//  NT: non-temporal stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float vf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void bt4(const float* d, float* o) { o[0] = d[0] - d[2]; o[1] = d[1] + d[2]; o[2] = d[2] - d[1]; o[3] = d[1] - d[3]; }

template <bool NT>
__device__ __forceinline__ void st2(float* q, float a, float b) {
    if (NT) { vf2 v; v.x = a; v.y = b; __builtin_nontemporal_store(v, reinterpret_cast<vf2*>(q)); }
    else *reinterpret_cast<float2*>(q) = make_float2(a, b);
}

template <bool NT>
__device__ __forceinline__ void emit(const float (&d)[4][6], float* o, size_t plane) {
    float v[4][4][2];
    #pragma unroll
    for (int q = 0; q < 2; ++q) {
        float r[4][4];
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float col[4] = {d[0][2 * q + j], d[1][2 * q + j], d[2][2 * q + j], d[3][2 * q + j]};
            float w[4]; bt4(col, w);
            r[0][j] = w[0]; r[1][j] = w[1]; r[2][j] = w[2]; r[3][j] = w[3];
        }
        #pragma unroll
        for (int i = 0; i < 4; ++i) { float w[4]; bt4(r[i], w); v[i][0][q] = w[0]; v[i][1][q] = w[1]; v[i][2][q] = w[2]; v[i][3][q] = w[3]; }
    }
    #pragma unroll
    for (int i = 0; i < 4; ++i)
        #pragma unroll
        for (int j = 0; j < 4; ++j) st2<NT>(o + (size_t)(4 * i + j) * plane, v[i][j][0], v[i][j][1]);
}

template <int VAR, bool NT, int R>
__global__ __launch_bounds__(256) void wino_in(const float* __restrict__ x, float* __restrict__ V, int N, int C, int H, int W) {
    const int TH = H / 2, TW = W / 2, TWP = TW / 2, THB = TH / R;
    const long long T = (long long)N * TH * TW;
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= (long long)N * THB * TWP) return;
    const int c = blockIdx.y;
    const int txp = (int)(u % TWP), tyb = (int)((u / TWP) % THB), n = (int)(u / ((long long)TWP * THB));
    const int tx = 2 * txp;
    const float* p = x + ((size_t)n * C + c) * H * W;
    const size_t plane = (size_t)C * T;
    const int x0 = 2 * tx - 1;
    const int lane = threadIdx.x & 63;
    float d[4][6];
    auto load_row = [&](int y, float (&row)[6]) {
        const bool yok = y >= 0 && y < H;
        const float* rp = p + (size_t)(yok ? y : 0) * W;
        float4 m = yok ? *reinterpret_cast<const float4*>(rp + x0 + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
        float e0 = 0.f, e5 = 0.f;
        if (VAR == 0) {
            e0 = (yok && x0 >= 0) ? rp[x0] : 0.f;
            e5 = (yok && x0 + 5 < W) ? rp[x0 + 5] : 0.f;
        } else if (VAR >= 3) {
            const float up = __shfl_up(m.w, 1), dn = __shfl_down(m.x, 1);
            e0 = txp == 0 ? 0.f : (lane == 0 ? (yok ? rp[x0] : 0.f) : up);
            e5 = txp == TWP - 1 ? 0.f : (lane == 63 ? (yok ? rp[x0 + 5] : 0.f) : dn);
        }
        row[0] = e0; row[1] = m.x; row[2] = m.y; row[3] = m.z; row[4] = m.w; row[5] = e5;
    };
    #pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ty = tyb * R + r;
        const int y0 = 2 * ty - 1;
        if (VAR == 2) {
            #pragma unroll
            for (int j = 0; j < 6; ++j) d[0][j] = d[3][j] = 0.f;
            load_row(y0 + 1, d[1]); load_row(y0 + 2, d[2]);
        } else if (r == 0 || VAR != 4) {
            #pragma unroll
            for (int i = 0; i < 4; ++i) load_row(y0 + i, d[i]);
        } else {
            #pragma unroll
            for (int j = 0; j < 6; ++j) { d[0][j] = d[2][j]; d[1][j] = d[3][j]; }
            load_row(y0 + 2, d[2]); load_row(y0 + 3, d[3]);
        }
        const size_t t = ((size_t)n * TH + ty) * TW + tx;
        emit<NT>(d, V + (size_t)c * T + t, plane);
    }
}

typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void st4(float* q, float a, float b, float c, float d) {
    if (NT) { vf4 v; v.x = a; v.y = b; v.z = c; v.w = d; __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(q)); }
    else *reinterpret_cast<float4*>(q) = make_float4(a, b, c, d);
}
// QUAD: 4 tiles per thread (8 columns = two float4 + halo via neighbour lanes), float4 stores (1 KB per wave and plane)
template <bool NT>
__global__ __launch_bounds__(256) void wino_in_quad(const float* __restrict__ x, float* __restrict__ V, int N, int C, int H, int W) {
    const int TH = H / 2, TW = W / 2, TWQ = TW / 4;
    const long long T = (long long)N * TH * TW;
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= (long long)N * TH * TWQ) return;
    const int c = blockIdx.y;
    const int txq = (int)(u % TWQ), ty = (int)((u / TWQ) % TH), n = (int)(u / ((long long)TWQ * TH));
    const int tx = 4 * txq;
    const float* p = x + ((size_t)n * C + c) * H * W;
    const size_t plane = (size_t)C * T;
    const int x0 = 2 * tx - 1, y0 = 2 * ty - 1;
    const int lane = threadIdx.x & 63;
    float d[4][10];
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = y0 + i;
        const bool yok = y >= 0 && y < H;
        const float* rp = p + (size_t)(yok ? y : 0) * W;
        const float4 a = yok ? *reinterpret_cast<const float4*>(rp + x0 + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b = yok ? *reinterpret_cast<const float4*>(rp + x0 + 5) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float up = __shfl_up(b.w, 1), dn = __shfl_down(a.x, 1);
        d[i][0] = txq == 0 ? 0.f : (lane == 0 ? (yok ? rp[x0] : 0.f) : up);
        d[i][9] = txq == TWQ - 1 ? 0.f : (lane == 63 ? (yok ? rp[x0 + 9] : 0.f) : dn);
        d[i][1] = a.x; d[i][2] = a.y; d[i][3] = a.z; d[i][4] = a.w; d[i][5] = b.x; d[i][6] = b.y; d[i][7] = b.z; d[i][8] = b.w;
    }
    float v[4][4][4];
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        float r[4][4];
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float col[4] = {d[0][2 * q + j], d[1][2 * q + j], d[2][2 * q + j], d[3][2 * q + j]};
            float w[4]; bt4(col, w);
            r[0][j] = w[0]; r[1][j] = w[1]; r[2][j] = w[2]; r[3][j] = w[3];
        }
        #pragma unroll
        for (int i = 0; i < 4; ++i) { float w[4]; bt4(r[i], w); v[i][0][q] = w[0]; v[i][1][q] = w[1]; v[i][2][q] = w[2]; v[i][3][q] = w[3]; }
    }
    float* o = V + (size_t)c * T + ((size_t)n * TH + ty) * TW + tx;
    #pragma unroll
    for (int i = 0; i < 4; ++i)
        #pragma unroll
        for (int j = 0; j < 4; ++j) st4<NT>(o + (size_t)(4 * i + j) * plane, v[i][j][0], v[i][j][1], v[i][j][2], v[i][j][3]);
}

int main() {
    const int N = 8, C = 256, H = 100, W = 168;
    const size_t nx = (size_t)N * C * H * W, T = (size_t)N * (H / 2) * (W / 2), nv = 16 * (size_t)C * T;
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
#define RUN(VAR, NT, R) run("VAR" #VAR " NT" #NT " R" #R, [&](float* x, float* v) { \
        const long long units = (long long)N * (H / 2 / R) * (W / 4); \
        wino_in<VAR, NT, R><<<dim3((unsigned)((units + 255) / 256), C), 256>>>(x, v, N, C, H, W); })
    RUN(0, false, 1); RUN(0, true, 1); RUN(1, false, 1); RUN(2, false, 1); RUN(2, true, 1); RUN(3, false, 1); RUN(3, true, 1);
    run("QUAD NTfalse", [&](float* x, float* v) { const long long units = (long long)N * (H / 2) * (W / 8);
        wino_in_quad<false><<<dim3((unsigned)((units + 255) / 256), C), 256>>>(x, v, N, C, H, W); });
    run("QUAD NTtrue", [&](float* x, float* v) { const long long units = (long long)N * (H / 2) * (W / 8);
        wino_in_quad<true><<<dim3((unsigned)((units + 255) / 256), C), 256>>>(x, v, N, C, H, W); });
    return 0;
}
