// FCOS ground-truth assignment for the whole mini-batch in ONE launch
//   [ref: models/customized_detectors/thirdparty_heads/fcos.py:177-284  FCOS.get_ground_truth]
// The reference builds, per image, (M x R x 4) ltrb tensors, an (M x R) centre-sampling mask per level, an (M x R) area
// matrix and an arg-min over it (R = 22,400 locations at 800x1344, M boxes) -- ~40 elementwise launches per image.  Here one
// thread owns one (image, location): it walks the image's boxes once, keeps the smallest-area candidate (first index on
// ties, like the arg-min) and emits class / ltrb deltas / centerness.  Every float expression is ONE IEEE operation per
// reference operation (contraction off, correctly rounded / and sqrt), so the targets are bit-identical to the elementwise
// definition.
#include <math.h>

#include "common.h"

#pragma clang fp contract(off)

namespace lgd {

struct FcosTargetArgs {
    const float* shifts;      // (R,2) location centres (x,y), levels concatenated
    const float* boxes;       // (T,4) GT boxes of all images, image-major
    const int64_t* classes;   // (T,)
    const int32_t* img_off;   // (B+1)
    int loc0[LGD_MAX_LEVELS + 1];   // first location of each level
    float lo[LGD_MAX_LEVELS], hi[LGD_MAX_LEVELS], rad[LGD_MAX_LEVELS];  // size range and centre-sampling radius in pixels
    int L, B, R, num_classes, center_sampling;
    int64_t* out_cls;         // (B,R)
    float* out_delta;         // (B,R,4) l,t,r,b
    float* out_ctr;           // (B,R)
};

__global__ __launch_bounds__(256) void fcos_target_kernel(FcosTargetArgs a) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (r >= a.R) return;
    int l = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) l += (i < a.L && r >= a.loc0[i]) ? 1 : 0;
    const float px = a.shifts[2 * r], py = a.shifts[2 * r + 1];
    const float lo = a.lo[l], hi = a.hi[l], rad = a.rad[l];
    const int m0 = a.img_off[b], m1 = a.img_off[b + 1];
    float best = INFINITY;
    int arg = -1;
    float bl = 0.f, bt = 0.f, br = 0.f, bb = 0.f;
    for (int m = m0; m < m1; ++m) {  // wave-uniform address: scalar loads
        const float x0 = a.boxes[4 * m], y0 = a.boxes[4 * m + 1], x1 = a.boxes[4 * m + 2], y1 = a.boxes[4 * m + 3];
        const float dl = px - x0, dt = py - y0, dr = x1 - px, db = y1 - py;
        bool inside;
        if (a.center_sampling) {
            const float cx = (x0 + x1) / 2.f, cy = (y0 + y1) / 2.f;
            const float sx0 = fmaxf(cx - rad, x0), sy0 = fmaxf(cy - rad, y0), sx1 = fminf(cx + rad, x1), sy1 = fminf(cy + rad, y1);
            inside = fminf(fminf(px - sx0, py - sy0), fminf(sx1 - px, sy1 - py)) > 0.f;
        } else {
            inside = fminf(fminf(dl, dt), fminf(dr, db)) > 0.f;
        }
        const float far = fmaxf(fmaxf(dl, dt), fmaxf(dr, db));
        const float area = (x1 - x0) * (y1 - y0);
        if (inside && far >= lo && far <= hi && area < best) { best = area; arg = m; bl = dl; bt = dt; br = dr; bb = db; }
    }
    const size_t o = (size_t)b * a.R + r;
    if (m1 == m0) {  // image without ground truth: background, zero targets
        a.out_cls[o] = a.num_classes;
        reinterpret_cast<float4*>(a.out_delta)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        a.out_ctr[o] = 0.f;
        return;
    }
    if (arg < 0) {  // no candidate: the arg-min over an all-inf column is index 0
        arg = m0;
        const float x0 = a.boxes[4 * m0], y0 = a.boxes[4 * m0 + 1], x1 = a.boxes[4 * m0 + 2], y1 = a.boxes[4 * m0 + 3];
        bl = px - x0; bt = py - y0; br = x1 - px; bb = y1 - py;
    }
    a.out_cls[o] = best == INFINITY ? (int64_t)a.num_classes : a.classes[arg];
    reinterpret_cast<float4*>(a.out_delta)[o] = make_float4(bl, bt, br, bb);
    // clamp_(min=0) keeps NaN (0/0 on a zero-extent box), as torch does; such locations are never foreground
    // division and square root through fp64: (float)((double)a / b) and (float)sqrt((double)x) ARE the correctly rounded fp32
    // results (53 >= 2*24 + 2 bits), whatever the compiler's fp32 division / sqrt expansion does
    float q0 = (float)((double)fminf(bl, br) / (double)fmaxf(bl, br)), q1 = (float)((double)fminf(bt, bb) / (double)fmaxf(bt, bb));
    q0 = q0 < 0.f ? 0.f : q0;
    q1 = q1 < 0.f ? 0.f : q1;
    a.out_ctr[o] = (float)sqrt((double)(q0 * q1));
}

}  // namespace lgd

extern "C" int lgd_fcos_targets(const float* shifts, const int32_t* level_locs_host, const float* size_lo_host,
                                const float* size_hi_host, const float* radius_px_host, int L, int R, const float* gt_boxes,
                                const int64_t* gt_classes, const int32_t* img_off, int B, int T, int num_classes,
                                int center_sampling, int64_t* out_classes, float* out_deltas, float* out_centerness,
                                void* stream) {
    if (!shifts || !level_locs_host || !size_lo_host || !size_hi_host || !radius_px_host || L < 1 || L > LGD_MAX_LEVELS || R < 1 ||
        B < 1 || T < 0 || !img_off || (T > 0 && (!gt_boxes || !gt_classes)) || !out_classes || !out_deltas || !out_centerness)
        return LGD_EINVAL;
    lgd::FcosTargetArgs a;
    int off = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.loc0[l] = off;
        a.lo[l] = a.hi[l] = a.rad[l] = 0.f;
        if (l < L) {
            if (level_locs_host[l] < 0) return LGD_EINVAL;
            off += level_locs_host[l];
            a.lo[l] = size_lo_host[l]; a.hi[l] = size_hi_host[l]; a.rad[l] = radius_px_host[l];
        }
    }
    a.loc0[LGD_MAX_LEVELS] = off;
    if (off != R) return LGD_EINVAL;
    a.shifts = shifts; a.boxes = gt_boxes; a.classes = gt_classes; a.img_off = img_off;
    a.L = L; a.B = B; a.R = R; a.num_classes = num_classes; a.center_sampling = center_sampling ? 1 : 0;
    a.out_cls = out_classes; a.out_delta = out_deltas; a.out_ctr = out_centerness;
    LGD_LAUNCH("fcos_target_kernel", lgd::fcos_target_kernel, dim3((R + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}
