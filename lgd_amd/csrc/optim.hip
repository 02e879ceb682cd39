// Gradient clipping + SGD(momentum, weight decay) of BOTH optimizers of the distillation step as ONE launch over a table of
// tensors [ref: train.py:200-204 -- stu_optimizer.step(); tea_optimizer.step() after detectron2's per-parameter gradient
// clipping (utils/build.py:514-529 maybe_add_gradient_clipping, CLIP_TYPE "value") and utils/build.py:494-512 torch.optim.SGD].
// torch's multi-tensor path runs clamp_min, clamp_max, g + wd p, mu buf, buf + g, p - lr buf as six passes per optimizer (13 tensor
// transfers per parameter element, ~40 launches) behind ~1.7 ms of Python per step (tensor grouping); at 2 images per GPU the GPU
// ran dry for 0.7 ms at the optimizer entry (tools/gap_profile.sh).  Here: one pass (read p, g, buf; write p, g, buf), one launch,
// the table is three device pointers + length + (lr, wd, mu) per tensor.
//   g   <- clamp(g, -clip, clip)            (written back: the clipped gradient is what a caller sees after the step)
//   d   <- g + wd p
//   buf <- mu buf + d                       (a zero-initialised buffer reproduces torch's first step, buf = d, exactly)
//   p   <- p - lr buf
#include "common.h"

#pragma clang fp contract(off)   // mu * buf + d keeps torch's two roundings (mul_ then add_); the fma calls below are explicit

namespace lgd {

constexpr int kSgdChunk = 4096;  // elements per workgroup: 256 threads x 4 float4

// 4 floats at a 4-byte-aligned address: under DistributedDataParallel(gradient_as_bucket_view=True) a gradient is a view into a flat
// bucket at the sum of the preceding parameters' sizes (train.py:279-281), i.e. dword- but not 16-byte-aligned as soon as one of them
// has a size that is not a multiple of 4 (FCOS `scales`, the centerness bias).  gfx950 runs global memory in unaligned access mode, so
// the accesses stay dwordx4 (the compiler emits them for this type); round 2 sent such tensors down the scalar path.
typedef float sgd_f4u __attribute__((ext_vector_type(4), aligned(4)));

template <bool VEC>
__device__ __forceinline__ void sgd_chunk(const lgd_sgd_tensor& t, long long base, float clip) {
    const float lr = t.lr, wd = t.wd, mu = t.mu;
    if constexpr (VEC) {
        float4 p[4], g[4], m[4];
        bool on[4];
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = base + (long long)(k * 256 + threadIdx.x) * 4;
            on[k] = i + 3 < t.n;
            if (on[k]) {
                p[k] = *reinterpret_cast<const float4*>(t.p + i); m[k] = *reinterpret_cast<const float4*>(t.m + i);
                const sgd_f4u gv = *reinterpret_cast<const sgd_f4u*>(t.g + i);
                g[k] = make_float4(gv.x, gv.y, gv.z, gv.w);
            }
        }
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!on[k]) continue;
            const long long i = base + (long long)(k * 256 + threadIdx.x) * 4;
            float* pp = &p[k].x; float* gg = &g[k].x; float* mm = &m[k].x;
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c = gg[j] < -clip ? -clip : (gg[j] > clip ? clip : gg[j]);   // NaN stays NaN, as torch.clamp
                const float d = __builtin_fmaf(wd, pp[j], c);
                const float b = mu * mm[j] + d;
                gg[j] = c; mm[j] = b; pp[j] = __builtin_fmaf(-lr, b, pp[j]);
            }
            *reinterpret_cast<float4*>(t.p + i) = p[k];
            sgd_f4u go; go.x = g[k].x; go.y = g[k].y; go.z = g[k].z; go.w = g[k].w;
            *reinterpret_cast<sgd_f4u*>(t.g + i) = go;
            *reinterpret_cast<float4*>(t.m + i) = m[k];
        }
    }
    // scalar path: parameters / momentum buffers that are not 16-byte aligned (never torch's own allocations), and the < 4-element
    // tail of a vector chunk
    const long long end = base + kSgdChunk < t.n ? base + kSgdChunk : t.n;
    long long i0 = base;
    if constexpr (VEC) i0 = end == t.n ? (t.n & ~3LL) : end;   // only the last chunk has a tail
    for (long long i = i0 + threadIdx.x; i < end; i += 256) {
        const float gi = t.g[i];
        const float c = gi < -clip ? -clip : (gi > clip ? clip : gi);
        const float d = __builtin_fmaf(wd, t.p[i], c);
        const float b = mu * t.m[i] + d;
        t.g[i] = c; t.m[i] = b; t.p[i] = __builtin_fmaf(-lr, b, t.p[i]);
    }
}

__global__ __launch_bounds__(256) void sgd_clip_kernel(const lgd_sgd_tensor* __restrict__ tab, const int32_t* __restrict__ blk_off,
                                                        int nt, float clip) {
    // the tensor this workgroup works on: last i with blk_off[i] <= blockIdx.x (wave-uniform: scalar loads)
    int lo = 0, hi = nt;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blk_off[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const lgd_sgd_tensor t = tab[lo];
    const long long base = (long long)((int)blockIdx.x - blk_off[lo]) * kSgdChunk;
    if (base >= t.n) return;
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.m)) & 15) == 0 && (reinterpret_cast<uintptr_t>(t.g) & 3) == 0;
    if (vec) sgd_chunk<true>(t, base, clip);
    else sgd_chunk<false>(t, base, clip);
}

}  // namespace lgd

extern "C" {

int lgd_sgd_chunk_elems(void) { return lgd::kSgdChunk; }

int lgd_sgd_clip_step(const lgd_sgd_tensor* table, const int32_t* blk_off, int n_tensors, int n_blocks, float clip_value, void* stream) {
    if (!table || !blk_off || n_tensors < 1 || n_blocks < 1 || !(clip_value > 0.f)) return LGD_EINVAL;
    LGD_LAUNCH("sgd_clip_kernel", lgd::sgd_clip_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, table, blk_off,
               n_tensors, clip_value);
    return lgd::check_launch();
}

}  // extern "C"
