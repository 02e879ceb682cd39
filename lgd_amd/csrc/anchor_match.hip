// Anchor <-> ground-truth matching of the student's detection loss, run twice per iteration on the same targets
// [ref: distillator.py:96-112 -> student.losses; detectron2 RetinaNet.label_anchors / Matcher:
//  IoU(gt, anchors), thresholds [0.4, 0.5] -> labels [0, -1, 1], allow_low_quality_matches, background -> K, ignore -> -1].
// The torch restatement is ~15 small kernels per image over (n_gt x R) matrices (R = 201,600 anchors at 800x1344); here
// two launches cover the whole mini-batch and never materialise the IoU matrix:
//   pass 1: best[g] = max over anchors of IoU(g, anchor)          (integer atomicMax on the IoU bits: order independent)
//   pass 2: per anchor max / first argmax over its image's boxes, thresholds, low-quality rule (IoU == best[g]), class
//           and matched box.
// The IoU is evaluated with the restatement's exact fp32 operation order (correctly rounded mul / add / div, no fma
// contraction), so the integer labels are bit-identical to the torch path.
#include "common.h"

#pragma clang fp contract(off)  // the IoU must round like the separate elementwise fp32 ops of its definition

namespace lgd {

struct MatchArgs {
    const float* anchors;     // (R, 4) x1,y1,x2,y2
    const float* gt;          // (T, 4)
    const int32_t* img_off;   // (B+1)
    const int64_t* cls;       // (T)
    unsigned* best;           // (T) IoU bits
    int64_t* labels;          // (B, R)
    float* matched;           // (B, R, 4)
    int R, B, K, low_quality;
    float lo, hi;
};

// plain operators, compiled under the file-scope contract(off): the header intrinsics (__fmul_rn, ...) are inline functions
// built with the default fp-contract=fast and DO get fused into v_fma after inlining (seen in the ISA)
__device__ __forceinline__ float box_area(const float4 b) { return (b.z - b.x) * (b.w - b.y); }

__device__ __forceinline__ float iou_xyxy(const float4 g, const float area_g, const float4 a, const float area_a) {
    const float w = fmaxf(fminf(g.z, a.z) - fmaxf(g.x, a.x), 0.f);
    const float h = fmaxf(fminf(g.w, a.w) - fmaxf(g.y, a.y), 0.f);
    const float inter = w * h;
    const float sum = area_g + area_a;
    const float den = sum - inter;
    return inter > 0.f ? inter / den : 0.f;  // '/' is the correctly rounded division (hipcc default)
}

__global__ __launch_bounds__(256) void anchor_best_kernel(MatchArgs m) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int g0 = m.img_off[b], g1 = m.img_off[b + 1];
    const bool on = r < m.R;
    const float4 a = on ? reinterpret_cast<const float4*>(m.anchors)[r] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float area_a = box_area(a);
    __shared__ float red[4];
    for (int g = g0; g < g1; ++g) {
        const float4 q = reinterpret_cast<const float4*>(m.gt)[g];
        float v = on ? iou_xyxy(q, box_area(q), a, area_a) : 0.f;
        v = wave_max(v);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float w = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (w > 0.f) atomicMax(m.best + g, __float_as_uint(w));  // IoU >= 0: the bit pattern orders like the value
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void anchor_label_kernel(MatchArgs m) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m.R) return;
    const int g0 = m.img_off[b], g1 = m.img_off[b + 1];
    const size_t o = (size_t)b * m.R + r;
    float4* mb = reinterpret_cast<float4*>(m.matched) + o;
    if (g1 == g0) {  // image without ground truth: everything background
        m.labels[o] = m.K;
        *mb = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 a = reinterpret_cast<const float4*>(m.anchors)[r];
    const float area_a = box_area(a);
    float vmax = -1.f;
    int imax = g0;
    bool lowq = false;
    for (int g = g0; g < g1; ++g) {
        const float4 q = reinterpret_cast<const float4*>(m.gt)[g];
        const float v = iou_xyxy(q, box_area(q), a, area_a);
        if (v > vmax) { vmax = v; imax = g; }  // first maximal index, as torch.max
        lowq |= (__float_as_uint(v) == m.best[g]);
    }
    int lab = vmax >= m.hi ? 1 : (vmax >= m.lo ? -1 : 0);
    if (m.low_quality && lowq) lab = 1;
    m.labels[o] = lab == 0 ? (int64_t)m.K : (lab < 0 ? (int64_t)-1 : m.cls[imax]);
    *mb = reinterpret_cast<const float4*>(m.gt)[imax];
}

}  // namespace lgd

extern "C" {

int lgd_anchor_match(const float* anchors, int R, const float* gt_boxes, const int64_t* gt_classes, const int32_t* img_off,
                     int B, int T, float iou_lo, float iou_hi, int num_classes, int allow_low_quality, uint32_t* best_ws,
                     int64_t* labels, float* matched_boxes, void* stream) {
    if (!anchors || !img_off || !labels || !matched_boxes || R < 1 || B < 1 || T < 0 || (T > 0 && (!gt_boxes || !gt_classes || !best_ws)))
        return LGD_EINVAL;
    lgd::MatchArgs m{anchors, gt_boxes, img_off, gt_classes, best_ws, labels, matched_boxes, R, B, num_classes,
                     allow_low_quality ? 1 : 0, iou_lo, iou_hi};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((R + 255) / 256, B), block(256);
    if (T > 0) {
        if (hipMemsetAsync(best_ws, 0, (size_t)T * sizeof(uint32_t), st) != hipSuccess) return LGD_ELAUNCH;
        LGD_LAUNCH("anchor_best_kernel", lgd::anchor_best_kernel, grid, block, 0, st, m);
    }
    LGD_LAUNCH("anchor_label_kernel", lgd::anchor_label_kernel, grid, block, 0, st, m);
    return lgd::check_launch();
}

}  // extern "C"
