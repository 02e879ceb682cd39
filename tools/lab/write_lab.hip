// HBM write ceiling: pure fill and 1:3 read:write streams over 1.2 GB (rotating buffers), float4 per lane, plain vs NT stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));

template <bool NT, int RD>   // RD: read one float4 per RD stores (0 = pure fill)
__global__ __launch_bounds__(256) void fill(const float* __restrict__ src, float* __restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    vf4 v; v.x = 1.f; v.y = 2.f; v.z = 3.f; v.w = 4.f;
    if (RD) { if (i % RD == 0) v = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(src) + i / RD); else v.x = (float)i; }
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(dst) + i);
    else reinterpret_cast<vf4*>(dst)[i] = v;
}

int main() {
    const size_t n = (size_t)300 * 1024 * 1024 / 4 * 4;  // 1.2 GB of floats? no: 300M floats = 1.2 GB
    const size_t n4 = n / 4;
    const int NB = 2;
    std::vector<float*> D(NB), S(NB);
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&D[i], n * 4)); CK(hipMalloc(&S[i], n * 4)); CK(hipMemset(S[i], 1, n * 4)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 2; ++i) launch(S[i % NB], D[i % NB]);
        CK(hipDeviceSynchronize());
        float tot = 0;
        for (int i = 0; i < 10; ++i) {
            CK(hipEventRecord(e0)); launch(S[i % NB], D[i % NB]); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
        }
        printf("%-40s avg %7.1f us  %.2f TB/s\n", name, tot / 10 * 1e3, bytes / (tot / 10 * 1e-3) / 1e12);
    };
    const unsigned blocks = (unsigned)((n4 + 255) / 256);
    run("fill plain stores", n * 4.0, [&](float* s, float* d) { fill<false, 0><<<blocks, 256>>>(s, d, n4); });
    run("fill NT stores", n * 4.0, [&](float* s, float* d) { fill<true, 0><<<blocks, 256>>>(s, d, n4); });
    run("read 1 : write 3, NT", n * 4.0 * (1 + 1.0 / 3), [&](float* s, float* d) { fill<true, 3><<<blocks, 256>>>(s, d, n4); });
    run("read 1 : write 1 (copy), NT", n * 4.0 * 2, [&](float* s, float* d) { fill<true, 1><<<blocks, 256>>>(s, d, n4); });
    run("read 1 : write 1 (copy), plain", n * 4.0 * 2, [&](float* s, float* d) { fill<false, 1><<<blocks, 256>>>(s, d, n4); });
    CK(hipMemsetAsync(D[0], 0, n * 4));
    run("hipMemsetAsync", n * 4.0, [&](float* s, float* d) { CK(hipMemsetAsync(d, 0, n * 4)); });
    return 0;
}
