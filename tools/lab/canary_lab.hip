// LAB (round 6): a canary kernel to run beside csrc/h2.hip's forward product on a second stream.  tools/conv_stage_probe.py showed that wino6_out --
// 256 threads, 16 KB of static LDS, ~122 VGPRs -- returns wrong values for the frequency plane it parks FIRST in LDS whenever h2_fwd (or gemm2h)
// runs beside it, also when that kernel touches no memory at all (LGD_H2_ABL=5), but not without its MFMAs.  What is it that changes under the
// canary: its registers, or its LDS?  Every thread holds NR known values in VGPRs (pinned by empty asm), the workgroup holds a known pattern in
// 16 KB of LDS; both are re-checked `iters` times with a sleep in between; mismatches are counted per kind and the first one is recorded.
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int NR = 80;

extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void canary_kernel(unsigned* out, int iters, int sleeps) {
    __shared__ uint32_t lds[4096];
    const unsigned t = threadIdx.x, salt = blockIdx.x * 2654435761u;
    uint32_t r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { r[i] = (t * 131u + i * 7919u) ^ salt; asm volatile("" : "+v"(r[i])); }
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k * 256 + t] = (k * 256 + t) * 2246822519u ^ salt;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int s = 0; s < sleeps; ++s) __builtin_amdgcn_s_sleep(64);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            asm volatile("" : "+v"(r[i]));
            if (r[i] != ((t * 131u + i * 7919u) ^ salt)) {
                if (atomicAdd(out + 0, 1u) == 0) { out[4] = i; out[5] = t; out[6] = blockIdx.x; out[7] = r[i]; out[8] = (t * 131u + i * 7919u) ^ salt; }
                r[i] = (t * 131u + i * 7919u) ^ salt;
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t v = lds[k * 256 + t], want = (k * 256 + t) * 2246822519u ^ salt;
            if (v != want) {
                if (atomicAdd(out + 1, 1u) == 0) { out[9] = k * 256 + t; out[10] = blockIdx.x; out[11] = v; out[12] = want; }
                lds[k * 256 + t] = want;
            }
        }
        __syncthreads();
    }
}

extern "C" int canary_launch(unsigned* out, int blocks, int iters, int sleeps, void* stream) {
    hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, sleeps);
    return (int)hipGetLastError();
}
