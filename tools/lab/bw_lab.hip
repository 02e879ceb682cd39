// Read-bandwidth lab: which streaming-read structure reaches the most of HBM on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ inline float4 ntload(const float4* p) { vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U, bool NT, bool F64>
__global__ __launch_bounds__(256) void chunk_reduce(const float* __restrict__ x, size_t n, int chunk, double* out) {
    // one wave per chunk (like in_moments): U float4 loads in flight per lane
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t e0 = (size_t)w * chunk;
    if (e0 >= n) return;
    const size_t e1 = min(n, e0 + (size_t)chunk);
    float s = 0.f; double d = 0;
    for (size_t e = e0 + lane * 4; e < e1; e += 256 * U) {
        float4 v[U];
        #pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t ee = e + u * 256;
            if (ee < e1) v[u] = NT ? ntload(reinterpret_cast<const float4*>(x + ee)) : *reinterpret_cast<const float4*>(x + ee);
            else v[u] = make_float4(0, 0, 0, 0);
        }
        #pragma unroll
        for (int u = 0; u < U; ++u) {
            if (F64) { d += (double)v[u].x; d = fma((double)v[u].y, (double)v[u].y, d); d += (double)v[u].z; d = fma((double)v[u].w, (double)v[u].w, d); }
            else s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
        }
    }
    if (F64) s = (float)d;
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[w] = s;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void grid_stride_reduce(const float* __restrict__ x, size_t n4, double* out) {
    // persistent: each thread strides over the whole array
    const size_t tid = blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    const float4* p = reinterpret_cast<const float4*>(x);
    float s = 0.f;
    size_t i = tid;
    for (; i + (U - 1) * nt < n4; i += U * nt) {
        float4 v[U];
        #pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? ntload(p + i + u * nt) : p[i + u * nt];
        #pragma unroll
        for (int u = 0; u < U; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    for (; i < n4; i += nt) { float4 v = p[i]; s += (v.x + v.y) + (v.z + v.w); }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}

template <int U>
__global__ __launch_bounds__(256) void block_contig_reduce(const float* __restrict__ x, size_t n4, size_t per_block4, double* out) {
    // each block owns a contiguous slab; threads stride by 256 inside it
    const float4* p = reinterpret_cast<const float4*>(x) + (size_t)blockIdx.x * per_block4;
    const size_t lim = min(per_block4, n4 - min(n4, (size_t)blockIdx.x * per_block4));
    float s = 0.f;
    size_t i = threadIdx.x;
    for (; i + (U - 1) * 256 < lim; i += U * 256) {
        float4 v[U];
        #pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
        #pragma unroll
        for (int u = 0; u < U; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    for (; i < lim; i += 256) { float4 v = p[i]; s += (v.x + v.y) + (v.z + v.w); }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}

int main() {
    const size_t n = (size_t)8 * 256 * 22400 * 2;  // 2 pyramids = 367 MB
    const int NB = 3;
    std::vector<float*> bufs(NB);
    for (auto& b : bufs) { CK(hipMalloc(&b, n * 4)); CK(hipMemset(b, 1, n * 4)); }
    double* out; CK(hipMalloc(&out, 1 << 24));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch(bufs[i % NB]);
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(a)); launch(bufs[i % NB]); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); best = fminf(best, ms); tot += ms;
        }
        printf("%-44s avg %7.1f us  %6.0f GB/s   best %7.1f us %6.0f GB/s\n", name, tot / 12 * 1e3, n * 4 / (tot / 12 * 1e-3) / 1e9, best * 1e3, n * 4 / (best * 1e-3) / 1e9);
    };
    const size_t n4 = n / 4;
#define CH(U, NT, F64, CHUNK) run("chunk U=" #U " NT=" #NT " F64=" #F64 " chunk=" #CHUNK, [&](float* x) { int waves = (n + CHUNK - 1) / CHUNK; chunk_reduce<U, NT, F64><<<(waves + 3) / 4, 256>>>(x, n, CHUNK, out); });
    CH(4, false, false, 4096) CH(4, true, false, 4096) CH(4, false, true, 4096) CH(8, false, false, 8192) CH(8, true, false, 8192) CH(8, false, false, 16384) CH(8, true, true, 16384) CH(16, false, false, 16384)
#define GS(U, NT, BLK) run("gridstride U=" #U " NT=" #NT " blocks=" #BLK, [&](float* x) { grid_stride_reduce<U, NT><<<BLK, 256>>>(x, n4, out); });
    GS(4, false, 2048) GS(8, false, 2048) GS(8, true, 2048) GS(8, false, 4096) GS(4, false, 8192) GS(16, false, 1024)
#define BC(U, BLK) run("blockcontig U=" #U " blocks=" #BLK, [&](float* x) { size_t pb = (n4 + BLK - 1) / BLK; block_contig_reduce<U><<<BLK, 256>>>(x, n4, pb, out); });
    BC(8, 2048) BC(8, 4096) BC(8, 16384) BC(4, 65536)
    // copy for reference
    run("hipMemcpyAsync D2D (read+write)", [&](float* x) { CK(hipMemcpyAsync(bufs[(x == bufs[0]) ? 1 : 0], x, n * 4 / 2, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
