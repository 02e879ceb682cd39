// access-shape lab: wave-per-plane, row-by-row (W*4 bytes per wave-load, lanes >= W/4 idle) vs flat 1 KB loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ inline float4 ld(const float* p) { vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }

// MODE 0: lane owns 4 columns, one row per load (42 lanes active for W=168), G rows in flight
// MODE 1: flat: wave reads the plane as 1 KB pieces (all lanes), G pieces in flight
// MODE 2: like 0 but TWO planes per wave interleaved (2G loads in flight)
template <int MODE, int G, int WPB>
__global__ __launch_bounds__(64 * WPB) void rows(const float* __restrict__ x, int nplanes, int H, int W, float* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int plane = blockIdx.x * WPB + wave;
    if (plane >= nplanes) return;
    const float* src = x + (size_t)plane * H * W;
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (MODE == 0) {
        const int xl = lane * 4;
        const bool on = xl < W;
        const float* col = src + (on ? xl : 0);
        for (int y0 = 0; y0 < H; y0 += G) {
            float4 v[G];
            #pragma unroll
            for (int u = 0; u < G; ++u) v[u] = ld(col + (size_t)min(y0 + u, H - 1) * W);
            #pragma unroll
            for (int u = 0; u < G; ++u) if (on && y0 + u < H) { s0 += v[u].x; s1 += v[u].y; s2 += v[u].z; s3 += v[u].w; }
        }
    } else {
        const int n = H * W;
        for (int e0 = 0; e0 < n; e0 += 256 * G) {
            float4 v[G];
            #pragma unroll
            for (int u = 0; u < G; ++u) { const int e = e0 + u * 256 + lane * 4; v[u] = ld(src + min(e, n - 4)); }
            #pragma unroll
            for (int u = 0; u < G; ++u) if (e0 + u * 256 + lane * 4 < n) { s0 += v[u].x; s1 += v[u].y; s2 += v[u].z; s3 += v[u].w; }
        }
    }
    float s = (s0 + s1) + (s2 + s3);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[plane] = s;
}

int main() {
    const int H = 100, W = 168, nplanes = 2048 + 512;  // p3 planes + p4-equivalent bytes
    const size_t n = (size_t)nplanes * H * W;
    const int NB = 6;
    std::vector<float*> X(NB);
    for (auto& b : X) { CK(hipMalloc(&b, n * 4)); CK(hipMemset(b, 1, n * 4)); }
    float* out; CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch(X[i % NB]);
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e0)); launch(X[i % NB]); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms); tot += ms;
        }
        printf("%-34s avg %7.1f us  %6.0f GB/s   best %7.1f us %6.0f GB/s\n", name, tot / 12 * 1e3, n * 4 / (tot / 12 * 1e-3) / 1e9, best * 1e3, n * 4 / (best * 1e-3) / 1e9);
    };
#define R(MODE, G, WPB) run("MODE=" #MODE " G=" #G " waves/blk=" #WPB, [&](float* x) { rows<MODE, G, WPB><<<(nplanes + WPB - 1) / WPB, 64 * WPB>>>(x, nplanes, H, W, out); });
    R(0, 4, 4) R(0, 8, 4) R(0, 16, 4) R(0, 8, 1) R(0, 8, 2) R(1, 4, 4) R(1, 8, 4) R(1, 8, 1)
    return 0;
}
