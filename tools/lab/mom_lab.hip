// in_moments structure lab: two streams, 5 fp64 moments per plane chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ inline float4 ld(const float* p) {
    if (NT) { vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *reinterpret_cast<const float4*>(p);
}
__device__ inline double wsum(double v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }

template <int U, bool NT, int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void mom(const float* __restrict__ a, const float* __restrict__ b, size_t n, int chunk, double* out) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t e0 = (size_t)w * chunk;
    if (e0 >= n) return;
    const size_t e1 = min(n, e0 + (size_t)chunk);
    double sa = 0, saa = 0, sb = 0, sbb = 0, sab = 0;
    const float pa0 = a[e0], pb0 = b[e0];
    for (size_t e = e0 + lane * 4; e < e1; e += 256 * U) {
        float4 va[U], vb[U];
        #pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t ee = e + u * 256;
            if (ee < e1) { va[u] = ld<NT>(a + ee); vb[u] = ld<NT>(b + ee); } else { va[u] = make_float4(pa0, pa0, pa0, pa0); vb[u] = make_float4(pb0, pb0, pb0, pb0); }
        }
        #pragma unroll
        for (int u = 0; u < U; ++u) {
            const float xa[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, xb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
            if (MODE == 0) {
                #pragma unroll
                for (int j = 0; j < 4; ++j) { const double x = xa[j], y = xb[j]; sa += x; saa = fma(x, x, saa); sb += y; sbb = fma(y, y, sbb); sab = fma(x, y, sab); }
            } else {  // shifted fp32 group sums, fp64 across groups
                float ga = 0, gaa = 0, gb = 0, gbb = 0, gab = 0;
                #pragma unroll
                for (int j = 0; j < 4; ++j) { const float x = xa[j] - pa0, y = xb[j] - pb0; ga += x; gaa = fmaf(x, x, gaa); gb += y; gbb = fmaf(y, y, gbb); gab = fmaf(x, y, gab); }
                sa += ga; saa += gaa; sb += gb; sbb += gbb; sab += gab;
            }
        }
    }
    sa = wsum(sa); saa = wsum(saa); sb = wsum(sb); sbb = wsum(sbb); sab = wsum(sab);
    if (lane == 0) { double* o = out + (size_t)w * 5; o[0] = sa; o[1] = saa; o[2] = sb; o[3] = sbb; o[4] = sab; }
}

int main() {
    const size_t n = (size_t)8 * 256 * 22400;
    const int NB = 4;
    std::vector<float*> A(NB), B(NB);
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&A[i], n * 4)); CK(hipMalloc(&B[i], n * 4)); CK(hipMemset(A[i], 1, n * 4)); CK(hipMemset(B[i], 2, n * 4)); }
    double* out; CK(hipMalloc(&out, 1 << 24));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch(A[i % NB], B[i % NB]);
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e0)); launch(A[i % NB], B[i % NB]); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms); tot += ms;
        }
        printf("%-40s avg %7.1f us  %6.0f GB/s   best %7.1f us %6.0f GB/s\n", name, tot / 12 * 1e3, 2 * n * 4 / (tot / 12 * 1e-3) / 1e9, best * 1e3, 2 * n * 4 / (best * 1e-3) / 1e9);
    };
#define M(U, NT, MODE, OCC, CHUNK) run("U=" #U " NT=" #NT " MODE=" #MODE " OCC=" #OCC " chunk=" #CHUNK, [&](float* a, float* b) { int waves = (n + CHUNK - 1) / CHUNK; mom<U, NT, MODE, OCC><<<(waves + 3) / 4, 256>>>(a, b, n, CHUNK, out); });
    M(4, false, 0, 1, 4096) M(4, true, 0, 1, 4096) M(2, false, 0, 1, 4096) M(2, true, 0, 1, 4096) M(1, true, 0, 1, 4096) M(2, true, 0, 2, 4096) M(2, true, 0, 1, 2048) M(1, true, 0, 2, 2048)
    M(4, true, 1, 1, 4096) M(2, true, 1, 1, 4096) M(2, true, 1, 2, 4096) M(1, true, 1, 2, 4096) M(2, true, 1, 2, 8192) M(2, false, 1, 2, 4096)
    return 0;
}
