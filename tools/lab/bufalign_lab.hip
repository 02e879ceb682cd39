// does buffer_load_dwordx4 honour a 4-byte-aligned (not 16-byte-aligned) offset on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* out) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 64 * 4, 0x00020000);
    const int off = threadIdx.x * 4;   // bytes
    u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));   // (without the bit_cast the conversion splats element 0)
    reinterpret_cast<u32x4*>(out)[threadIdx.x] = v;   // (stored whole: indexing the elements made hipcc emit a dword load and a splat)
}
int main() {
    float h[64], *d, *o, r[64 * 4];
    for (int i = 0; i < 64; ++i) h[i] = (float)i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o); hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int t = 0; t < 8; ++t) printf("lane %d (offset %d B): %g %g %g %g\n", t, t * 4, r[t * 4], r[t * 4 + 1], r[t * 4 + 2], r[t * 4 + 3]);
    for (int t = 60; t < 64; ++t) printf("lane %d (offset %d B): %g %g %g %g\n", t, t * 4, r[t * 4], r[t * 4 + 1], r[t * 4 + 2], r[t * 4 + 3]);
    return 0;
}
