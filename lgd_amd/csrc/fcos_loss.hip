// FCOS box-regression (GIoU) and centerness (BCE) losses on the head's RAW NCHW outputs, with the head's own epilogue folded in
// (runs twice per iteration: student features and teacher features).
//   [ref: thirdparty_heads/fcos.py:533-546  bbox_pred = scale_l(bbox_pred(tower)); relu(.) * stride_l (NORM_REG_TARGETS) or exp(.);
//         thirdparty_heads/fcos.py:107-175  losses(): iou_loss(pred[fg], gt[fg], centerness_targets, box_mode="ltrb", "giou", "sum")
//         / max(1, sum of centerness targets over ranks), binary_cross_entropy_with_logits(centerness[fg], targets, "sum") / num_fg]
// The composed form runs, per head pass, three elementwise kernels and their backward per map for the Scale / ReLU / stride epilogue
// (10 maps), permute + cat copies of every regression and centerness map to (N, R, .), and ~45 elementwise kernels for the boolean-free
// GIoU / BCE and their autograd: ~360 launches of 5-15 us per step at config 3.  Here ONE kernel walks the locations: for foreground
// locations only it applies scale_l / ReLU / stride_l to the 4 raw regression values read in place, evaluates the GIoU loss against
// the target (weighted by the centerness target) and the centerness BCE, and -- the two normalisers are known when the loss is
// evaluated and the upstream gradient of a loss term in the training step is 1 -- writes the gradients w.r.t. the RAW regression map,
// the centerness logits (dense: zeros elsewhere) and, per block, the partial gradient of the level's scale in the same pass.
// Sums in fp64 block partials, fixed-order reduction (bit-reproducible).
#include "common.h"

namespace lgd {

struct FcosLossArgs {
    const float* reg[LGD_MAX_LEVELS];     // (N, 4, H, W) raw bbox_pred output
    const float* ctr[LGD_MAX_LEVELS];     // (N, 1, H, W) centerness logits
    float* greg[LGD_MAX_LEVELS];
    float* gctr[LGD_MAX_LEVELS];
    int HW[LGD_MAX_LEVELS], r0[LGD_MAX_LEVELS], blk0[LGD_MAX_LEVELS + 1];
    float stride[LGD_MAX_LEVELS];
    const float* scales;                  // [L] the per-level Scale parameters
    const long long* cls;                 // (N, R) int64: foreground iff 0 <= c < K
    const float* tgt;                     // (N, R, 4) ltrb targets
    const float* tctr;                    // (N, R) centerness targets
    const float* norm;                    // [2] 1 / num_targets (box), 1 / num_foreground (centerness)
    double* ws;                           // [nblk][3] box sum, centerness sum, d scale partial
    float* out;                           // [2 + L] loss_box, loss_centerness, d scales
    int L, N, K, R, nblk, norm_reg;
};

__device__ __forceinline__ float fl_softplus_neg_abs(float x) {   // log1p(exp(-|x|))
    const float e = __expf(-fabsf(x));
    return e < 1e-4f ? e * (1.f - 0.5f * e) : __logf(1.f + e);
}

__global__ __launch_bounds__(256) void fcos_loss_kernel(FcosLossArgs a) {
    int l = 0;
    #pragma unroll
    for (int i = 1; i < LGD_MAX_LEVELS; ++i) l += (i < a.L && (int)blockIdx.x >= a.blk0[i]) ? 1 : 0;
    const int HW = a.HW[l];
    const long long i = (long long)(blockIdx.x - a.blk0[l]) * 256 + threadIdx.x;   // over N * HW
    double sb = 0.0, sc = 0.0, ss = 0.0;
    if (i < (long long)a.N * HW) {
        const int e = (int)(i % HW), n = (int)(i / HW);
        const size_t r = (size_t)n * a.R + a.r0[l] + e;
        const long long c = a.cls[r];
        float g[4] = {0.f, 0.f, 0.f, 0.f}, gc = 0.f;
        if (c >= 0 && c < a.K) {
            const float s = a.scales[l], st = a.stride[l], eps = 1.1920929e-07f;
            const float4 t = reinterpret_cast<const float4*>(a.tgt)[r];
            const float w = a.tctr[r];
            const float* pz = a.reg[l] + (size_t)n * 4 * HW + e;
            float z[4], d[4], dd[4];   // raw value, decoded distance, d distance / d (z * s)
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                z[j] = pz[(size_t)j * HW];
                const float u = z[j] * s;
                if (a.norm_reg) { d[j] = fmaxf(u, 0.f) * st; dd[j] = u > 0.f ? st : 0.f; }
                else { d[j] = __expf(u); dd[j] = d[j]; }
            }
            // cvpods iou_loss(box_mode="ltrb", loss_type="giou"): boxes (-l, -t, r, b) around the location
            const float tg[4] = {t.x, t.y, t.z, t.w};
            const float pw = d[2] + d[0], ph = d[3] + d[1];
            const float pa = fmaxf(pw, 0.f) * fmaxf(ph, 0.f);
            const float ta = fmaxf(tg[2] + tg[0], 0.f) * fmaxf(tg[3] + tg[1], 0.f);
            const float aw = fminf(d[2], tg[2]) + fminf(d[0], tg[0]), ah = fminf(d[3], tg[3]) + fminf(d[1], tg[1]);
            const float wi = fmaxf(aw, 0.f), hi = fmaxf(ah, 0.f);
            const float inter = wi * hi, uni = ta + pa - inter, U = fmaxf(uni, eps);
            const float gw = fmaxf(d[2], tg[2]) + fmaxf(d[0], tg[0]), gh = fmaxf(d[3], tg[3]) + fmaxf(d[1], tg[1]);
            const float ac = gw * gh, A = fmaxf(ac, eps);
            const float loss = 1.f - (inter / U - (ac - uni) / A);
            sb = (double)(loss * w);
            // reverse mode
            const float G_uni = (uni >= eps ? inter / (U * U) : 0.f) - 1.f / A;   // d loss / d union (torch.clamp passes the gradient on its bound)
            const float G_inter = -1.f / U - G_uni;                              // union = ta + pa - inter
            const float G_ac = ac >= eps ? uni / (A * A) : 1.f / A;
            const float G_aw = aw >= 0.f ? G_inter * hi : 0.f, G_ah = ah >= 0.f ? G_inter * wi : 0.f;
            const float G_pw = pw >= 0.f ? G_uni * fmaxf(ph, 0.f) : 0.f, G_ph = ph >= 0.f ? G_uni * fmaxf(pw, 0.f) : 0.f;
            const float G_gw = G_ac * gh, G_gh = G_ac * gw;
            // torch.min / torch.max hand the gradient to the first operand where it is strictly smaller / larger, half of it on a tie
            auto lt = [](float x, float y) { return x < y ? 1.f : (x == y ? 0.5f : 0.f); };
            const float Gd[4] = {G_aw * lt(d[0], tg[0]) + G_pw + G_gw * lt(tg[0], d[0]),
                                 G_ah * lt(d[1], tg[1]) + G_ph + G_gh * lt(tg[1], d[1]),
                                 G_aw * lt(d[2], tg[2]) + G_pw + G_gw * lt(tg[2], d[2]),
                                 G_ah * lt(d[3], tg[3]) + G_ph + G_gh * lt(tg[3], d[3])};
            const float kb = w * a.norm[0];
            float dsc = 0.f;
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gu = kb * Gd[j] * dd[j];      // d loss_box / d (z * s)
                g[j] = gu * s;
                dsc += gu * z[j];
            }
            ss = (double)dsc;
            // centerness: BCE with logits against the target
            const float x = a.ctr[l][(size_t)n * HW + e];
            sc = (double)(fmaxf(x, 0.f) - x * w + fl_softplus_neg_abs(x));
            const float ex = __expf(-fabsf(x)), sg = x >= 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex);
            gc = (sg - w) * a.norm[1];
        }
        float* og = a.greg[l] + (size_t)n * 4 * HW + e;
        #pragma unroll
        for (int j = 0; j < 4; ++j) og[(size_t)j * HW] = g[j];
        a.gctr[l][(size_t)n * HW + e] = gc;
    }
    __shared__ double red[4][3];
    sb = wave_sum(sb); sc = wave_sum(sc); ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sb; red[threadIdx.x >> 6][1] = sc; red[threadIdx.x >> 6][2] = ss; }
    __syncthreads();
    if (threadIdx.x < 3) a.ws[(size_t)blockIdx.x * 3 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// item 0: box loss, item 1: centerness loss, items 2 ..: d scale of level (item - 2); item i is summed by wave i % 4 with a fixed
// assignment of partials to lanes
__global__ __launch_bounds__(256) void fcos_loss_reduce_kernel(FcosLossArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int item = wave; item < 2 + a.L; item += 4) {
        int b0 = 0, b1 = a.nblk, col = item;
        if (item >= 2) { b0 = a.blk0[item - 2]; b1 = a.blk0[item - 1]; col = 2; }
        double s = 0.0;
        for (int i = b0 + lane; i < b1; i += 64) s += a.ws[(size_t)i * 3 + col];
        s = wave_sum(s);
        if (lane == 0) a.out[item] = (float)(item == 0 ? s * (double)a.norm[0] : (item == 1 ? s * (double)a.norm[1] : s));
    }
}

static int fcos_loss_fill(FcosLossArgs& a, const float* const* reg, const float* const* ctr, const int32_t* level_hw, const float* strides,
                          int L, int N, int K, int R) {
    if (!reg || !ctr || !level_hw || !strides || L < 1 || L > LGD_MAX_LEVELS || N < 1 || K < 1) return LGD_EINVAL;
    a.L = L; a.N = N; a.K = K; a.R = R;
    int r0 = 0, blk = 0;
    for (int l = 0; l < LGD_MAX_LEVELS; ++l) {
        a.reg[l] = nullptr; a.ctr[l] = nullptr; a.greg[l] = nullptr; a.gctr[l] = nullptr; a.HW[l] = 0; a.r0[l] = 0; a.stride[l] = 1.f;
        a.blk0[l] = blk;
        if (l < L) {
            if (!reg[l] || !ctr[l]) return LGD_EINVAL;
            a.reg[l] = reg[l]; a.ctr[l] = ctr[l];
            a.HW[l] = level_hw[2 * l] * level_hw[2 * l + 1];
            a.stride[l] = strides[l];
            a.r0[l] = r0;
            r0 += a.HW[l];
            blk += (int)(((long long)N * a.HW[l] + 255) / 256);
        }
    }
    if (r0 != R) return LGD_EINVAL;
    a.blk0[LGD_MAX_LEVELS] = blk;
    for (int l = L; l < LGD_MAX_LEVELS; ++l) a.blk0[l] = blk;
    a.nblk = blk;
    return LGD_OK;
}

}  // namespace lgd

extern "C" {

size_t lgd_fcos_loss_ws_doubles(const int32_t* level_hw_host, int L, int N) {
    size_t blk = 0;
    for (int l = 0; l < L; ++l) blk += ((size_t)N * level_hw_host[2 * l] * level_hw_host[2 * l + 1] + 255) / 256;
    return 3 * blk;
}

int lgd_fcos_loss_fwd_grad(const float* const* reg_host, const float* const* ctr_host, const int32_t* level_hw_host,
                           const float* strides_host, int L, int N, int K, int R, const float* scales, int norm_reg_targets,
                           const long long* gt_classes, const float* gt_deltas, const float* gt_centerness, const float* inv_norm2,
                           double* ws, float* out, float* const* grad_reg_host, float* const* grad_ctr_host, void* stream) {
    lgd::FcosLossArgs a;
    if (lgd::fcos_loss_fill(a, reg_host, ctr_host, level_hw_host, strides_host, L, N, K, R) != LGD_OK || !scales || !gt_classes ||
        !gt_deltas || !gt_centerness || !inv_norm2 || !ws || !out || !grad_reg_host || !grad_ctr_host) return LGD_EINVAL;
    for (int l = 0; l < L; ++l) {
        if (!grad_reg_host[l] || !grad_ctr_host[l]) return LGD_EINVAL;
        a.greg[l] = grad_reg_host[l]; a.gctr[l] = grad_ctr_host[l];
    }
    a.scales = scales; a.cls = gt_classes; a.tgt = gt_deltas; a.tctr = gt_centerness; a.norm = inv_norm2; a.ws = ws; a.out = out;
    a.norm_reg = norm_reg_targets ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    LGD_LAUNCH("fcos_loss_kernel", lgd::fcos_loss_kernel, dim3(a.nblk), dim3(256), 0, s, a);
    LGD_LAUNCH("fcos_loss_reduce_kernel", lgd::fcos_loss_reduce_kernel, dim3(1), dim3(256), 0, s, a);
    return lgd::check_launch();
}

}  // extern "C"
