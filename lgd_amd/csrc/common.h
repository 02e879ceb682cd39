// Shared device helpers for the gfx950 LGD kernels (wave64 only; no portability shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lgd_hip.h"

#define LGD_WAVE 64

namespace lgd {

// ---- DPP cross-lane (no LDS traffic). ctrl: quad_perm 0x00-0xFF, row_mirror 0x140, row_half_mirror 0x141.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float readlane_f32(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// neighbour lane's value across the whole wave in ONE VALU op (DPP wave_shr:1 / wave_shl:1) instead of an LDS ds_bpermute:
// shr1: lane i <- lane i-1 (lane 0 reads 0); shl1: lane i <- lane i+1 (lane 63 reads 0)
__device__ __forceinline__ float wave_shr1(float v) { return dpp_f32<0x138>(v); }
__device__ __forceinline__ float wave_shl1(float v) { return dpp_f32<0x130>(v); }
__device__ __forceinline__ unsigned wave_shr1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned wave_shl1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }

// Sum over the 64 lanes of a wave; result is wave-uniform. Fixed order => deterministic.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);  // row_half_mirror: 8-lane sums
    v += dpp_f32<0x140>(v);  // row_mirror: 16-lane (row) sums in every lane
    return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}

// ---- running maximum of non-negative floats, kept as their bit pattern in ONE device word (the magnitude bounds of csrc/h2.hip).  Tens of
// thousands of waves report to the same word: an atomic each serialises in the L2 (measured: 530 us for a 25 us pass over the maps).  Nearly
// all of them lose against the value already there, which a relaxed load settles; a stale (smaller) value only costs an unnecessary atomic.
__device__ __forceinline__ void atomic_max_bits(unsigned* addr, unsigned v) {
    if (v > __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(addr, v);
}
// the same from a 256-thread workgroup: `am` >= 0 is wave-uniform (a wave_max); `slots` = 4 floats of LDS nobody else touches until the next
// barrier the caller executes.  ONE load (and rarely an atomic) per workgroup: even the loads serialise at ~1 per clock on their L2 line
// (measured on the F(6x6) output transform: 45 K of them cost +19 us per launch).
__device__ __forceinline__ void block_max_bits(unsigned* addr, float am, float* slots) {
    if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) atomic_max_bits(addr, __builtin_bit_cast(unsigned, fmaxf(fmaxf(slots[0], slots[1]), fmaxf(slots[2], slots[3]))));
}

// ---- streaming (non-temporal) 16-byte load: data that is read exactly once should not displace L2 lines.
// Measured on MI355X (tools/lab/bw_lab.hip): +11 % read bandwidth over plain loads for one-pass reductions.
typedef float lgd_vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg_stream4(const float* p) {
    const lgd_vf4 v = __builtin_nontemporal_load(reinterpret_cast<const lgd_vf4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ldg_stream(const float* p) { return __builtin_nontemporal_load(p); }

// ---- vector load/store of VW consecutive floats (VW in {1,2,4}); address must be VW*4-byte aligned.
template <int VW> struct Vec;
template <> struct Vec<1> { float v[1]; };
template <> struct Vec<2> { float v[2]; };
template <> struct Vec<4> { float v[4]; };

template <int VW>
__device__ __forceinline__ Vec<VW> vload(const float* p) {
    Vec<VW> r;
    if constexpr (VW == 4) { const float4 t = ldg_stream4(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
    else if constexpr (VW == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
    else { r.v[0] = *p; }
    return r;
}
template <int VW>
__device__ __forceinline__ void vstore(float* p, const Vec<VW>& r) {
    if constexpr (VW == 4) { *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]); }
    else if constexpr (VW == 2) { *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1]); }
    else { *p = r.v[0]; }
}

// ---- bf16x3 split of fp32 values (csrc/gemm3.hip and the filter transforms that emit its operand image): x = h + m + l with
// h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), round to nearest even; packed two elements per dword, the first in the low half
typedef float lgd_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 lgd_bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lgd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const lgd_bf16x2 v = __builtin_convertvector((lgd_f32x2){a, b}, lgd_bf16x2);   // v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {   // 11 VALU operations per pair
    h = pack_bf16(x0, x1);
    float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = pack_bf16(r0, r1);
    r0 -= __builtin_bit_cast(float, m << 16);
    r1 -= __builtin_bit_cast(float, m & 0xffff0000u);
    l = pack_bf16(r0, r1);
}
// eight consecutive-k values of one row -> the three 16-byte fragments of the gemm3 operand image
// [batch][k-step of 16][piece][32-row block][lane = (k % 16 / 8) * 32 + row % 32][8 bf16]; d: the h fragment, piece stride rbp KB
__device__ __forceinline__ void store_split8(const float* x, char* d, long rbp) {
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2(x[2 * e], x[2 * e + 1], h[e], m[e], l[e]);
    *reinterpret_cast<lgd_u32x4*>(d) = (lgd_u32x4){h[0], h[1], h[2], h[3]};
    *reinterpret_cast<lgd_u32x4*>(d + rbp * 1024) = (lgd_u32x4){m[0], m[1], m[2], m[3]};
    *reinterpret_cast<lgd_u32x4*>(d + 2 * rbp * 1024) = (lgd_u32x4){l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ long gemm3_image_off(long b, int ktp, int rbp, int m, int k) {   // byte offset of the h fragment holding (m, k)
    return ((((b * ktp + (k >> 4)) * 3) * rbp + (m >> 5)) << 10) + ((((k >> 3) & 1) * 32 + (m & 31)) << 4);
}

// ---- geometry workspace layout (int32 units): rects [L][B][max_n][4] | nbp [L][B] | bands [L][B][maxbp]
// (rectangles are padded per image so that a wave's start-up loads -- counts, band table, its box -- depend only on
//  (level, image) and issue together: one memory latency instead of a chain of three)
__host__ __device__ inline int geom_maxbp(int max_n) { return 2 * max_n + 2; }
__host__ __device__ inline size_t geom_rects_off() { return 0; }
__host__ __device__ inline size_t geom_nbp_off(int L, int B, int max_n) { return (size_t)L * B * max_n * 4; }
__host__ __device__ inline size_t geom_bands_off(int L, int B, int max_n) { return geom_nbp_off(L, B, max_n) + (size_t)L * B; }

// Thread-local record of the last launch failure (see lgd_last_error()).
void set_last_error(hipError_t e);
inline int check_launch() {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error(e); return LGD_ELAUNCH; }
    return LGD_OK;
}

// RAII event pair around one kernel launch (timing.hip); a no-op unless lgd_timing_enable(1).
class KTimer {
public:
    KTimer(const char* name, hipStream_t s);
    ~KTimer();
private:
    const char* name_; hipStream_t s_; hipEvent_t a_, b_;
};

}  // namespace lgd

#define LGD_LAUNCH(name, kernel, grid, block, smem, stream, ...)                 \
    do {                                                                          \
        lgd::KTimer _lgd_t(name, stream);                                         \
        hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);       \
    } while (0)
