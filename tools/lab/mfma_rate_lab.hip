// what v_mfma_f32_32x32x16_f16 sustains on this part: CH independent accumulator chains per wave, WPS waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    if (s == 123.456f) out[0] = s;
}
template <int CH> void run(int wgs_per_cu, float* d) {
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CH><<<grid, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<CH><<<grid, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)grid * 4 * iters * CH, fl = mf * 32768.0;
    printf("chains %d, %d wave(s) per SIMD: %.3f ms, %.0f TFLOP/s, %.1f ns per MFMA per SIMD\n", CH, wgs_per_cu, ms, fl / ms * 1e-9, ms * 1e6 / ((double)iters * CH * wgs_per_cu));
}
int main() {
    float* d; hipMalloc(&d, 4);
    run<1>(1, d); run<2>(1, d); run<4>(1, d); run<8>(1, d); run<1>(2, d); run<4>(2, d); run<8>(2, d);
    return 0;
}
