/*
 * lgd_hip.h -- C ABI of the MI355X (gfx950) LGD hot-path kernels.
 *
 * The reference (megvii-research/LGD) is pure Python over detectron2 and has no
 * FFI of its own; this header is the drop-in boundary the
 * reference's Python call sites bind through ctypes.  Every
 * entry point cites the reference code it replaces as
 *     [ref: <path under /root/reference>:<lines>].
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`
 *   - feature maps are fp32, NCHW, contiguous: (B, C, H_l, W_l) per FPN level l
 *   - boxes of the whole mini-batch are concatenated image-major: T = sum_b N_b rows,
 *     img_off[b] .. img_off[b+1] are image b's rows (int32, device, B+1 entries);
 *     with ADD_CONTEXT_BOX the context box is the LAST row of each image
 *   - per-level (T, C) tables are stored level-major: [L][T][C]
 *   - `stream` is a hipStream_t passed as void* (0 = default stream)
 *   - functions never allocate, never synchronise, never throw; they return 0 on
 *     success or a negative LGD_E* code.  Workspaces are caller-provided.
 */
#ifndef LGD_HIP_H
#define LGD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* maps per call: 5 pyramid levels, or the 2 x 5 maps of the single head pass over student + teacher features */
#define LGD_MAX_LEVELS 16

#define LGD_OK 0
#define LGD_EINVAL (-1)   /* bad argument (null pointer, L > LGD_MAX_LEVELS, C % 4 != 0, ...) */
#define LGD_ELAUNCH (-2)  /* hipLaunchKernel reported an error */

/* ABI version, bumped on any signature change. */
int lgd_abi_version(void);
/* Name of the GPU arch the library was compiled for ("gfx950"). */
const char* lgd_arch(void);
/* hipGetLastError() as text for the calling thread's last failed launch ("" if none). */
const char* lgd_last_error(void);

/* ------------------------------------------------------------------ box geometry
 * Integer pixel rectangles + row bands for every (level, box).
 * [ref: models/customized_detectors/dynamic_teacher/utils.py:53-89  get_inside_gt_mask]
 * The reference materialises a dense (N_b, H*W) float mask per level and image; the mask
 * is separable and its per-axis true-set is one interval, so it is carried as
 * rect = [x0, x1, y0, y1] (inclusive; empty box -> x1 < x0).  The inclusion predicate
 * |centre - p| / extent <= 0.5 is evaluated in fp32 with the reference's exact operation
 * order for every pixel coordinate p (bit-exact masks).
 *
 * boxes    : (T,4) fp32 x1,y1,x2,y2 in padded-image pixels, already clamped
 *            [ref: label_encoder.py:83-85 -- the `boxlists` values]
 * geom     : int32 workspace of lgd_geom_ints(L,B,T,max_n) entries, filled here and
 *            consumed by the box kernels below.
 */
size_t lgd_geom_ints(int L, int B, int T, int max_n);
int lgd_box_prep(const float* boxes, const int32_t* img_off, int B, int T, int max_n,
                 int img_h, int img_w, const int32_t* level_hw_host /* L x (H,W) */, int L,
                 int32_t* geom, void* stream);
/* Offsets (in int32 units) of the sub-tables inside `geom`, for tests/debugging:
 * rects [L][B][max_n][4] (padded per image), nbp [L][B], bands [L][B][2*max_n+2]. */
size_t lgd_geom_rects_off(int L, int B, int T, int max_n);
size_t lgd_geom_nbp_off(int L, int B, int T, int max_n);
size_t lgd_geom_bands_off(int L, int B, int T, int max_n);

/* ------------------------------------------------------------------ K1/K3: box reduce / box paint
 * box_sum : out[l][t][c] = sum over pixels inside box t of feat_l[b(t)][c][y][x]
 *           (normalize=1: divided by max(pixel count, 1))
 *   forward of the appearance encoder (mask pooling)
 *     [ref: dynamic_teacher.py:81-103 aggregate_per_level]
 *   and backward of the rendering w.r.t. the projected embeddings (normalize=0).
 * box_paint: out_l[b][c][y][x] = sum over boxes t of image b covering (y,x) of vals[l][t][c]
 *           (normalize=1: each box's value is first divided by max(pixel count, 1))
 *   forward of the intra-object knowledge mapping's scatter
 *     [ref: dynamic_teacher.py:137,173  torch.mm(attn_output.T, inside_mask)]
 *   and backward of mask pooling w.r.t. the feature map (normalize=1).
 * skip_last=1 treats the last box of every image as empty (the context box is not
 *   painted [ref: dynamic_teacher.py:123,131]).
 * feats_host / outs_host: host arrays of L device pointers.
 */
size_t lgd_box_pool_ws_floats(const int32_t* level_hw_host, int L, int B, int C, int max_n,
                              int outputs /* 1: lgd_box_sum, 2: lgd_gn_pool_fwd */);
int lgd_box_sum(const float* const* feats_host, const int32_t* level_hw_host, int L, int B, int C, int T,
                int max_n, const int32_t* img_off, const int32_t* geom,
                float* ws /* lgd_box_pool_ws_floats(..., 1) floats: per-chunk partial sums */, float* out /* [L][T][C] */,
                int normalize, int skip_last, void* stream);
int lgd_box_paint(const float* vals /* [L][T][C] */, const int32_t* level_hw_host, int L, int B, int C, int T,
                  int max_n, const int32_t* img_off, const int32_t* geom, float* const* outs_host,
                  int normalize, int skip_last, void* stream);

/* ------------------------------------------------------------------ K5+K1 fused: pool(relu(GroupNorm1(x)))
 * The appearance encoder pools `student_proj_2D(feat)` = conv3x3 -> GN(1) -> ReLU  [ref: dynamic_teacher.py:57,235,
 * 249-253] and nothing else reads that map, so the normalised map is never written: after lgd_gn1 statistics
 * (gn_stats [L*B][2] mean, rstd -- e.g. from lgd_gn1_fwd's stats pass), lgd_gn_pool_fwd streams the CONV OUTPUT x
 * once, applies (x-mean)*rstd and ReLU in registers and accumulates the box means: out [L][T][C]; next to them it
 * keeps raw [2][L][T][C]: per (box, channel) the sum of relu(xhat) and the number of pixels with xhat > 0
 * (ws: lgd_box_pool_ws_floats(..., 2) floats).
 * lgd_gn_pool_bwd: dx for the conv output from dpool [L][T][C]; the painted gradient is composed per row band on
 * the fly (never materialised); the GroupNorm backward's two means come from dpool and raw (no pass over x);
 * bstats = [L*B][2] scratch.
 * HBM traffic: fwd P (+P for the statistics), bwd P read + P write  (unfused: 4P / 6P).
 */
int lgd_gn_pool_fwd(const float* const* x_host, const float* gn_stats, const int32_t* level_hw_host, int L, int B,
                    int C, int T, int max_n, const int32_t* img_off, const int32_t* geom, float* ws, float* out,
                    float* raw, void* stream);
int lgd_gn_pool_bwd(const float* const* x_host, const float* gn_stats, const float* dpool, const float* raw,
                    const int32_t* level_hw_host, int L, int B, int C, int T, int max_n, const int32_t* img_off,
                    const int32_t* geom, float* bstats, float* const* dx_host, void* stream);

/* ------------------------------------------------------------------ K4: InstanceNorm x2 + MSE
 * [ref: models/base_distillator.py:59-64  norm_stu / norm_tea (InstanceNorm2d(256, affine=False),
 *  eps 1e-5, biased variance), flatten+cat over levels, coef * F.mse_loss]
 * loss = coef / (B*C*sum_l H_l*W_l) * sum_{l,b,c,hw} (IN(a)-IN(b))^2, computed in ONE pass
 * over a and b from per-plane fp64 moments (sum a, a^2, b, b^2, ab).
 *   ws     : fp64 workspace of lgd_distill_ws_doubles(...) entries
 *   stats  : fp32 [nplanes][8] per-(level,b,c) statistics kept for the backward
 *            (mean_a, rstd_a, mean_b, rstd_b, q, 0,0,0), nplanes = L*B*C
 *   loss   : 1 fp32
 * backward (student side only; the teacher side is detached in the reference,
 * base_distillator.py:55):  grad_a_l = gscale * dloss/da_l, gscale read from device
 * (upstream gradient, 1 fp32) times coef.
 */
size_t lgd_distill_ws_doubles(const int32_t* level_hw_host, int L, int B, int C);
int lgd_distill_fwd(const float* const* a_host, const float* const* b_host, const int32_t* level_hw_host,
                    int L, int B, int C, float coef, double* ws, float* stats, float* loss, void* stream);
int lgd_distill_bwd(const float* const* a_host, const float* const* b_host, const int32_t* level_hw_host,
                    int L, int B, int C, float coef, const float* stats, const float* grad_loss,
                    float* const* grad_a_host, void* stream);

/* ------------------------------------------------------------------ K5: GroupNorm(1 group, no affine) [+ReLU]
 * [ref: dynamic_teacher/layers.py:6-7 get_norm, 22-32 get_CONVS; dynamic_teacher.py:57 student_proj_2D,
 *  67-73 refinement_module]  y = (x - mean_b) * rsqrt(var_b + 1e-5) over each sample's C*H*W elements
 * (biased variance), optionally followed by ReLU; all pyramid levels in one call.
 *   ws    : fp64 workspace, lgd_gn1_ws_doubles(...) entries (shared by fwd and bwd)
 *   stats : fp32 [L*B][2] (mean, rstd), written by fwd, read by bwd
 *   bstats: fp32 [L*B][2] scratch of the backward (mean(g), mean(g*xhat))
 * backward: dx = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy * [y > 0] when relu.
 */
size_t lgd_gn1_ws_doubles(const int32_t* level_hw_host, int L, int B, int C);
int lgd_gn1_fwd(const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C, int relu,
                double* ws, float* stats, float* const* y_host, void* stream);
/* statistics pass of lgd_gn1_fwd alone (mean, rstd per (level, sample)); used by the fused lgd_gn_pool_* path */
int lgd_gn1_stats(const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C, double* ws,
                  float* stats, void* stream);
int lgd_gn1_bwd(const float* const* x_host, const float* const* dy_host, const int32_t* level_hw_host, int L,
                int B, int C, int relu, const float* stats, double* ws, float* bstats,
                float* const* dx_host, void* stream);

/* ------------------------------------------------------------------ GroupNorm(G groups, per-channel affine) [+ReLU] (FCOS towers)
 * [ref: models/customized_detectors/thirdparty_heads/fcos.py:455-470 (tower = conv3x3, GroupNorm(32, C), ReLU), 520-531
 *  (the towers are applied to every level)]  y = gamma[c] * (x - mean_{b,g}) * rsqrt(var_{b,g} + 1e-5) + beta[c] over each
 * sample's group of C/G adjacent channel planes (biased variance), optionally followed by ReLU; all maps in one call.
 * gamma / beta: [C] or NULL (1 / 0).
 *   ws        : fp64 workspace, lgd_gn_group_ws_doubles(...) entries (shared by fwd and bwd)
 *   stats     : fp32 [L*B*G][2] (mean, rstd), written by fwd, read by bwd
 *   bstats    : fp32 [L*B*G][2] scratch of the backward
 *   plane_sums: fp32 [L*B*C][2] per (map, sample, channel): sum g, sum g*xhat with g = dy * [y > 0] when relu;
 *               dbeta[c] / dgamma[c] are their sums over maps and samples (added by the host binding)
 * backward: dx = rstd * (gamma*g - mean_grp(gamma*g) - xhat * mean_grp(gamma*g*xhat)).
 */
size_t lgd_gn_group_ws_doubles(const int32_t* level_hw_host, int L, int B, int C);
int lgd_gn_group_fwd(const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C, int G, const float* gamma,
                     const float* beta, int relu, double* ws, float* stats, float* const* y_host, void* stream);
/* statistics of lgd_gn_group_fwd alone, folded with gamma / beta into affine [L][B][C][2] = (rstd * gamma, beta - mean * rstd * gamma):
 * what lgd_wino_in(pre_affine) applies while it loads (the normalised maps are never written) */
int lgd_gn_group_stats_affine(const float* const* x_host, const int32_t* level_hw_host, int L, int B, int C, int G,
                              const float* gamma, const float* beta, double* ws, float* stats, float* affine, void* stream);
int lgd_gn_group_bwd(const float* const* x_host, const float* const* dy_host, const int32_t* level_hw_host, int L, int B, int C,
                     int G, const float* gamma, const float* beta, int relu, const float* stats, double* ws, float* bstats,
                     float* plane_sums, float* const* dx_host, void* stream);
/* the statistics half of lgd_gn_group_bwd(relu = 0) alone: bstats, plane_sums as above and coef [L][B][C][4] = (ca, cm, mean, cb) with
 * dx = ca * g - cm - (x - mean) * cb, ca = rstd * gamma, cm = rstd * m1, cb = rstd^2 * m2 -- what lgd_wino_out_t_gn of the convolution
 * that PRODUCED x applies while it loads g and x: the backward apply pass of a conv -> GroupNorm -> ReLU -> conv tower layer
 * (thirdparty_heads/fcos.py:455-470) never runs and dx is never written or re-read. */
int lgd_gn_group_bwd_coef(const float* const* x_host, const float* const* dy_host, const int32_t* level_hw_host, int L, int B, int C,
                          int G, const float* gamma, const float* stats, double* ws, float* bstats, float* plane_sums, float* coef,
                          float* wmax, uint32_t* bound_out, void* stream);
/* (round 5) wmax (as many floats as lgd_gn_group_ws_doubles counts doubles) + bound_out (a word zeroed by the caller), both optional: max |dx| of
 * the gradient lgd_wino_out_t_gn(_h2) will form, bounded PER PLANE from the chunk maxima of |g| and |xhat| -- the f16x2 scale of that transform. */

/* ------------------------------------------------------------------ K3b: ReLU(x + ctx[b,c]) epilogue of the rendering
 * [ref: dynamic_teacher.py:151  F.relu(inst_featmap + ctx_feature[:, :, None, None])]
 * ctx / dctx: [L][B][C].  backward takes the saved OUTPUT y: dx = dy * [y > 0], dctx = sum_hw dx.
 */
int lgd_ctx_relu_fwd(const float* const* x_host, const float* ctx, const int32_t* level_hw_host, int L, int B,
                     int C, float* const* y_host, void* stream);
int lgd_ctx_relu_bwd(const float* const* y_host, const float* const* dy_host, const int32_t* level_hw_host,
                     int L, int B, int C, float* const* dx_host, float* dctx, void* stream);

/* ------------------------------------------------------------------ K2: block-diagonal multi-head cross-attention
 * [ref: dynamic_teacher.py:76-78 nn.MultiheadAttention(256, 8); 255-273 attn_mask + per-level calls]
 *
 * lgd_gemm_batch: up to 16 small fp32 GEMMs in one launch on the MFMA pipe (v_mfma_f32_16x16x4_f32),
 *   C[m,n] = alpha * (sum_k A(m,k) * B(n,k) + bias[n]),  A(m,k) = A[m*sa_m + k*sa_k], B(n,k) = B[n*sb_n + k*sb_k],
 *   C at C[m*sc_m + n*sc_n]; rowsum (optional) [M] = alpha * sum_k A(m,k)  (bias gradients).
 *   Used for the in/out projections [ref: torch MultiheadAttention in_proj_weight (3E,E) / out_proj] and
 *   their backward products.  The problem array lives in HOST memory.
 * lgd_attn_fwd: per (image, head) softmax(q k^T) v over the image's own boxes only (the reference's
 *   (T,T) boolean mask blocks exactly the cross-image pairs).  q is already scaled by 1/sqrt(E/H).
 *   q (Lq,T,E), k/v (Lk,T,E), Lq == Lk or one of them 1 (operand shared by all levels);
 *   out (L,T,E), lse (L,T,H) = log-sum-exp of each score row (kept for the backward), L = max(Lq,Lk).
 * lgd_attn_bwd: dq, dk, dv are PER-LEVEL partials, each (L,T,E); for an operand shared by all levels (Lq or
 *   Lk == 1) the caller sums its partials over the level axis.
 */
typedef struct lgd_gemm_problem {
    const float* A;
    const float* B;
    const float* bias;   /* may be NULL */
    float* C;
    float* rowsum;       /* may be NULL */
    int32_t M, N, K, reserved0;
    int64_t sa_m, sa_k, sb_n, sb_k, sc_m, sc_n;
    float alpha;
    int32_t reserved1;
} lgd_gemm_problem;
int lgd_gemm_batch(const lgd_gemm_problem* probs_host, int np, void* stream);
int lgd_attn_fwd(const float* q, const float* k, const float* v, const int32_t* img_off, int Lq, int Lk, int B,
                 int T, int E, int H, float* out, float* lse, void* stream);
int lgd_attn_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse,
                 const float* dout, const int32_t* img_off, int Lq, int Lk, int B, int T, int E, int H,
                 float* dq, float* dk, float* dv, void* stream);

/* ------------------------------------------------------------------ K6: label-encoder glue
 * lgd_box_descriptors [ref: label_encoder.py:12-115 box_descriptor_encode, utils.py:16-51]: per-box 4+K descriptor
 *   (clamped xyxy / image size, one-hot class, scaled to [-1,1]) and the clamped boxes ("boxlists"), on the device:
 *   boxes_in (T0,4) instance boxes as annotated (x1y1x2y2, or x1y1wh when wh_format), classes (T0,) int32,
 *   in_off/out_off (B+1) int32 row offsets of the instances / of the output rows (instances [+1 context row],
 *   or ONE substitute row [0,0,1,1] for an image without ground truth).  desc (T,4+K), boxes_out (T,4).
 * lgd_rowln_*   : LayerNorm over the last axis, no affine, eps 1e-5 [+ReLU]  [ref: label_encoder.py:157-160,243-270;
 *   spatial_transformer.py:23-40; layers.py:9-19]; stats (T,2) = mean, rstd.
 * lgd_rowvecmat_*: out[t,:] = x[t,:] @ M[t,:,:]  (T-Net transforms)  [ref: label_encoder.py:241,248]; k <= 128.
 * lgd_segmax_*  : per-image max over the image's rows, broadcast back to them  [ref: label_encoder.py:195-213,262-264];
 *   arg (B,F) int32 = arg-max row, consumed by the backward.
 */
int lgd_box_descriptors(const float* boxes_in, const int32_t* classes, const int32_t* in_off, const int32_t* out_off,
                        int B, int T, int num_classes, int img_h, int img_w, int add_ctx, int wh_format,
                        float* desc, float* boxes_out, void* stream);
int lgd_rowln_fwd(const float* x, int T, int F, int relu, float* y, float* stats, void* stream);
int lgd_rowln_bwd(const float* x, const float* dy, const float* stats, int T, int F, int relu, float* dx, void* stream);
int lgd_rowvecmat_fwd(const float* x, const float* M, int T, int k, float* out, void* stream);
int lgd_rowvecmat_bwd(const float* x, const float* M, const float* dout, int T, int k, float* dx, float* dM,
                      void* stream);
int lgd_segmax_fwd(const float* x, const int32_t* off, int B, int F, float* out, int32_t* arg, void* stream);
int lgd_segmax_bwd(const float* dout, const int32_t* off, const int32_t* arg, int B, int F, float* dx, void* stream);

/* ------------------------------------------------------------------ sigmoid focal loss on raw NCHW head outputs (section 8f-1)
 * [ref: distillator.py:107-112 / 288-295 -> student.losses -> fvcore sigmoid_focal_loss_jit(alpha, gamma, "sum");
 *  thirdparty_heads/fcos.py:146-152]  logits_l: (N, A*K, H_l, W_l) fp32 NCHW as the head's conv emits them;
 * labels_l: (N, A, H_l, W_l) int32, class index in [0,K), K = background, < 0 = ignored anchor.
 * loss = sum over non-ignored anchors and classes; backward writes grad_loss[0] * dloss/dlogits in NCHW.
 * lgd_focal_loss_fwd_grad: the sum AND grad_scale[0] * dsum/dlogits (grad_scale: device scalar, NULL = 1) in one pass over the
 * logits -- for a loss that is divided by a normaliser known when it is evaluated (detectron2's EMA of the positive count,
 * FCOS's foreground count) and whose upstream gradient in the step is 1; its backward is lgd_scale_unless_one(grads, upstream):
 * x_l *= g[0] for the L tensors unless g[0] == 1 (then every workgroup leaves after one load).
 * bound_out (may be NULL): receives the float bits of |grad_scale| max(alpha, 1 - alpha) (1 + gamma / e) >= max |gradient| -- the magnitude tag the
 * class convolution's backward derives its f16x2 scale from (csrc/h2.hip) without a pass over the gradient maps; lgd_scale_unless_one multiplies
 * *bound_inout (may be NULL) by |g[0]| along with the maps.
 */
size_t lgd_focal_ws_doubles(const int32_t* level_hw_host, int L, int N, int A, int K);
int lgd_focal_loss_fwd(const float* const* logits_host, const int32_t* const* labels_host,
                       const int32_t* level_hw_host, int L, int N, int A, int K, float alpha, float gamma,
                       double* ws, float* loss, void* stream);
int lgd_focal_loss_fwd_grad(const float* const* logits_host, const int32_t* const* labels_host,
                            const int32_t* level_hw_host, int L, int N, int A, int K, float alpha, float gamma,
                            const float* grad_scale, double* ws, float* loss, float* const* grad_logits_host, uint32_t* bound_out, void* stream);
int lgd_scale_unless_one(float* const* x_host, const long long* n_host, int L, const float* g, uint32_t* bound_inout, void* stream);
int lgd_focal_loss_bwd(const float* const* logits_host, const int32_t* const* labels_host,
                       const int32_t* level_hw_host, int L, int N, int A, int K, float alpha, float gamma,
                       const float* grad_loss, float* const* grad_logits_host, void* stream);

/* ------------------------------------------------------------------ FCOS regression (GIoU) + centerness (BCE) losses (section 8f-1)
 * [ref: thirdparty_heads/fcos.py:533-546 (per-level Scale, relu(.) * stride or exp(.)), 107-175 (losses)]
 * reg_l: (N, 4, H_l, W_l) RAW bbox_pred outputs, ctr_l: (N, 1, H_l, W_l) centerness logits; scales [L] (device), strides [L] (host);
 * gt_classes (N, R) int64 (foreground: 0 <= c < K), gt_deltas (N, R, 4) ltrb, gt_centerness (N, R), R = sum H_l W_l;
 * inv_norm2 (device): 1 / max(1, sum of centerness targets), 1 / max(1, number of foreground locations) [rank means].
 * out [2 + L] = loss_box_reg, loss_centerness (both normalised), d loss_box_reg / d scale_l; grad_reg_l / grad_ctr_l: the gradients
 * of the two normalised losses w.r.t. the raw maps (dense), written in the same pass (upstream gradient 1; rescale with
 * lgd_scale_unless_one otherwise).  ws: lgd_fcos_loss_ws_doubles(...) doubles.
 */
size_t lgd_fcos_loss_ws_doubles(const int32_t* level_hw_host, int L, int N);
int lgd_fcos_loss_fwd_grad(const float* const* reg_host, const float* const* ctr_host, const int32_t* level_hw_host,
                           const float* strides_host, int L, int N, int K, int R, const float* scales, int norm_reg_targets,
                           const long long* gt_classes, const float* gt_deltas, const float* gt_centerness, const float* inv_norm2,
                           double* ws, float* out, float* const* grad_reg_host, float* const* grad_ctr_host, void* stream);

/* ------------------------------------------------------------------ K8: 3x3 convolutions (Winograd data transforms)
 * Replaces the data movement of nn.Conv2d(C, C', 3, padding=1) on the path: dynamic_teacher.py:57,61,67-73
 * (student_proj_2D, local_inst_proj_2D, refinement_module), models/adapters/sequential_convs.py:10-12 and the
 * student head re-run on the teacher features (distillator.py:107-109).  Each of these modules applies ONE filter
 * to every pyramid level, so the tiles of all L levels are concatenated: T = lgd_wino_tiles(level_hw, L, N, tile) =
 * sum_l pad16(N * ceil(H_l/tile) * ceil(W_l/tile)) (every level is followed by all-zero pad tiles up to a multiple of 16).
 * tile = 4: F(4x4,3x3), nf = 36 frequencies, windows 6x6 at stride 4 (2.25 multiplies per output pixel, fp32 rounding ~1e-5 of the
 *   output scale per convolution); tile = 6: F(6x6,3x3), nf = 64, windows 8x8 at stride 6 (1.78 multiplies, ~1.8e-5).
 * The nf per-frequency channel products M[f] = U[f] (C' x C) @ V[f] (C x T) between the transforms are plain library
 * GEMMs issued by the host (rocBLAS / hipBLASLt); U = G g G^T comes from lgd_wino_filter_fwd.
 * V, M, dM, dV are [C][nf][T] fp32, tile index fastest: the GEMM of frequency f sees a (C x T) matrix with leading dimension nf*T
 * and batch stride T.  x_host / y_host / dy_host / dx_host: host arrays of L device pointers to (N, C, H_l, W_l) maps.
 *   lgd_wino_in   : V  = B^T d B  of the (tile+2)^2 input window of every tile (stride tile, zero halo 1)
 *   lgd_wino_out  : y  = A^T m A + bias (bias may be NULL) [then ReLU if relu], tile x tile outputs per tile, clipped to H x W
 *   lgd_wino_out_t: dM = A (dy . mask) A^T, the adjoint of lgd_wino_out: the ONE expansion of dy the backward needs
 *                   (dU[f] = dM[f] @ V[f]^T, dV[f] = U[f]^T @ dM[f])
 *   lgd_wino_in_t : dx = the adjoint of lgd_wino_in: overlap-add of B dV B^T over the windows, written as a gather per
 *                   tile x tile block (deterministic, no atomics)
 * Mask tables (relu_bits / pre_bits, may be NULL): [C][T] entries of lgd_wino_mask_bytes(tile) bytes (uint16 for tile 4, uint64 for
 *   tile 6), bit tile*i+j set <=> pixel (i,j) of the tile's own block is > 0.  lgd_wino_out writes relu_bits (with relu = 1);
 *   lgd_wino_out_t / lgd_wino_in_t_out_t take it as the gradient mask: 1 bit per pixel instead of re-reading the 4-byte forward
 *   output, and the forward output need not be kept for the backward.
 * pre_bias (may be NULL): the maps are PRE-activations of a conv -> per-channel bias -> ReLU whose output feeds this
 *   convolution and nothing else (conv1 -> FrozenBN -> ReLU -> conv2 of a bottleneck block): lgd_wino_in transforms
 *   relu(x + pre_bias[c]) -- the epilogue pass of the producing convolution is folded into this load -- and writes the activation
 *   mask per tile to pre_bits (may be NULL when no backward follows); lgd_wino_in_t takes pre_bits and returns the
 *   gradient of the RAW maps (zero where the activation was <= 0).
 * pre_affine (may be NULL; excludes pre_bias): [L][N][C][2] (scale, shift) per (map, sample, channel) -- the maps are the inputs
 *   of a GroupNorm + ReLU whose statistics are folded into scale = rstd * gamma, shift = beta - mean * rstd * gamma
 *   (lgd_gn_group_stats_affine; the FCOS towers' conv -> GroupNorm(32) -> ReLU -> conv, thirdparty_heads/fcos.py:455-470):
 *   lgd_wino_in transforms relu(x * scale + shift), the normalised map is never written; pre_bits as above -- lgd_wino_in_t then
 *   returns the gradient w.r.t. the GroupNorm OUTPUT (before the ReLU), which lgd_gn_group_bwd(relu = 0) takes. */
size_t lgd_wino_tiles(const int32_t* level_hw_host, int L, int N, int tile);
size_t lgd_wino_mask_bytes(int tile);
int lgd_wino_in(const float* const* x_host, const int32_t* level_hw_host, int L, int N, int C, int tile, float* V,
                const float* pre_bias, const float* pre_affine, void* pre_bits, void* stream);
int lgd_wino_out(const float* M, const float* bias, const int32_t* level_hw_host, int L, int N, int C, int tile,
                 int relu, float* const* y_host, void* relu_bits, void* stream);
int lgd_wino_out_t(const float* const* dy_host, const void* relu_bits, const int32_t* level_hw_host, int L, int N, int C, int tile,
                   float* dM, void* stream);
int lgd_wino_in_t(const float* dV, const int32_t* level_hw_host, int L, int N, int C, int tile, float* const* dx_host,
                  const void* pre_bits, void* stream);
/* lgd_wino_out_t of a convolution whose outputs y feed a GroupNorm (tile 6 only): g_host are the gradients w.r.t. the GroupNorm OUTPUT,
 * coef [L][N][C][4] comes from lgd_gn_group_bwd_coef; dM = A (ca * g - cm - (y - mean) * cb) A^T. */
int lgd_wino_out_t_gn(const float* const* g_host, const float* const* y_host, const float* coef, const int32_t* level_hw_host, int L,
                      int N, int C, int tile, float* dM, void* stream);
/* Filter transforms:
 *   lgd_wino_filter_fwd: U[f][co][ci] = (G (scale[co] . g) G^T)[f] at U + f*u_plane + co*Ci + ci, and the same values transposed in
 *     (co, ci) at Ut + f*ut_plane + ci*ut_ld + co (the operand of dV = U^T dM; Ut may be NULL: a caller whose GEMM takes a transposed
 *     operand needs U alone and the transform writes half the bytes); u_plane / ut_plane / ut_ld let several filters stack
 *     their slabs along C_out in one buffer (lgd_amd/ops.py::_Conv3x3K).  w: (Co, Ci, 3, 3) contiguous; scale: per-output-channel
 *     factor of a frozen affine that follows the convolution (detectron2 FrozenBatchNorm2d) or NULL.
 *   lgd_wino_filter_bwd: dw[co][ci] = scale[co] . G^T dU[:, co, ci] G, dU read at dU + f*du_plane + co*Ci + ci. */
int lgd_wino_filter_fwd(const float* w, const float* scale, int Co, int Ci, int tile, float* U, long long u_plane, float* Ut,
                        long long ut_ld, long long ut_plane, void* stream);
/* lgd_wino_filter_images (tile 6): the same transform written as the bf16x3 operand images of lgd_gemm3 instead of fp32 U -- img_fwd: image
 *   of U (64 batches, M = Ct rows, K = Ci) for M = U V; img_bwd: image of U^T (M = Ci, K = Ct) for dV = U^T dM; either may be NULL.  The
 *   filter's rows are row0 .. row0 + Co of a stack of Ct output channels (K filters on shared maps).  Co, Ci, row0, Ct multiples of 16. */
int lgd_wino_filter_images(const float* w, const float* scale, int Co, int Ci, int tile, int row0, int Ct, void* img_fwd, void* img_bwd,
                           void* stream);
int lgd_wino_filter_bwd(const float* dU, long long du_plane, const float* scale, int Co, int Ci, int tile, float* dw, void* stream);
/* lgd_wino_in_t followed by lgd_wino_out_t of the convolution that PRODUCED these maps, in one kernel: the backward link of a
 * conv -> [ReLU] -> conv chain whose intermediate maps have no other consumer (the head towers, the adapter: distillator.py:107-109,
 * sequential_convs.py:10-12).  dV: frequency-domain input gradient of the later conv; relu_bits: the earlier conv's mask table (NULL: no
 * ReLU between them); dM: A (dx . mask) A^T of the earlier conv, same [C][nf][T] layout.  The gradient map dx is never materialised. */
int lgd_wino_in_t_out_t(const float* dV, const void* relu_bits, const int32_t* level_hw_host, int L, int N, int C, int tile, float* dM,
                        void* stream);

/* ------------------------------------------------------------------ K9: channel products on the bf16 MFMA pipe, fp32 in and out
 * The per-frequency channel GEMMs of the Winograd convolutions  M[f] = U[f] V[f]  and  dV[f] = U[f]^T dM[f]
 * [the arithmetic of nn.Conv2d(C, C', 3, padding=1): dynamic_teacher.py:57,61,67-73,145,280; sequential_convs.py:10-12; the head towers
 *  distillator.py:107-109] as strided-batched products  C[b] (M x N) = A[b] (M x K) B[b] (K x N)  with fp32 operands and result, computed
 * on v_mfma_f32_32x32x16_bf16: every fp32 element x is split x = h + m + l into three bf16 pieces (round to nearest; h + m + l = x to
 * 2^-24 |x|) and 6 of the 9 cross products (ah bh, ah bm, am bh, ah bl, am bm, al bh) are accumulated in fp32: the dropped terms are
 * <= 2^-23 |a||b|, i.e. the result is an fp32-class product (measured error against fp64: 6.2e-7 of the output scale, rocBLAS fp32:
 * 7.6e-7) at 6/16 of the fp32 MFMA pipe's time.
 *   A is the FILTER operand: it is split ahead of the product (lgd_gemm3_split) into an image in MFMA fragment order,
 *   [batch][K/16][piece 3][ceil(M/32)][lane 64][8 bf16], lgd_gemm3_image_bytes(nb, M, K) bytes, which the product kernel copies into
 *   LDS by LDS-DMA; element (m, k) of batch b is read at A[b * a_sb + m * a_sm + k * a_sk] (any strides: U and U^T are both views).
 *   B (K x N) and C (M x N) have their last axis contiguous: B[b * b_sb + k * b_sk + n], C[b * c_sb + m * c_sm + n]; B is split inside
 *   the product kernel while it is staged (each element once: a workgroup's tile spans 256 rows of A).
 *   image_shared: ONE image (split with nb = 1) serves every batch -- the student's 1x1 convolutions, W (C' x C) against a batch of
 *   (C x HW) maps [d2-memory: BottleneckBlock conv1 / conv3, FPN laterals; SURVEY.md appendix A].  Tiles of 256 x 128 (C' = 128: 128 x 128).
 *   Epilogue (each part optional, NULL / 0 = absent; runs on the 128-row tile): C = relu?(A B + R + shift[m]) -- R (M x N, R[b * r_sb + m * r_sm
 *   + n]) the residual map, R == C with C's strides: C += A B (the input gradient of a block's first 1x1 convolution lands on the
 *   shortcut's gradient); shift[m] the frozen affine's per-channel shift; relu_bits: the ReLU mask, bit n % 32 of word
 *   (b * M + m) * ceil(N / 32) + n / 32 set <=> C(m, n) > 0 (lgd_relu_rowbits_words(nb * M, N) words; consumed by lgd_relu_rowbits_bwd)
 *   [d2-memory: BottleneckBlock.forward -- out = conv3(out); out += shortcut; out = relu(out)].
 * Requirements (LGD_EINVAL otherwise; the host falls back to the library GEMM): K % 16 == 0, 16-byte aligned image. */
size_t lgd_gemm3_image_bytes(int nb, int M, int K);
int lgd_gemm3_split(const float* A, long long a_sb, long long a_sm, long long a_sk, int nb, int M, int K, void* image, void* stream);
int lgd_gemm3(const void* image, int image_shared, const float* B, long long b_sb, long long b_sk, float* C, long long c_sb, long long c_sm,
              const float* R, long long r_sb, long long r_sm, const float* shift, int relu, uint32_t* relu_bits, uint32_t* amax_out, int nb, int M, int N,
              int K, void* stream);
/* (round 5) amax_out: optional word (zeroed by the caller) that receives max |C| as stored, as float bits: the magnitude bound of lgd_h2_*. */
/* lgd_gemm2h: the same product, tiles and epilogues in the f16x2 form of K10 -- the student's 1x1 convolutions (an fp32-class product: error against fp64 as
 * lgd_gemm3's): the filter as ONE two-piece f16 image for all batches (lgd_gemm2h_split: A 2^ea, ea from the bound *a_amax of |A|, 2^-ea recorded in
 * a_inv[0]), the activation operand scaled by the power of two its bound *b_amax (max |B|, float bits: the tag the producing kernel left, or
 * lgd_h2_amax_maps) prescribes and split in registers: three MFMAs per k-step instead of six, half the split arithmetic, two thirds of the LDS traffic. */
size_t lgd_gemm2h_image_bytes(int nb, int M, int K);
int lgd_gemm2h_split(const float* A, long long a_sb, long long a_sm, long long a_sk, int nb, int M, int K, const uint32_t* a_amax, void* image, float* a_inv,
                     void* stream);
/* lgd_gemm2h_split for a table of n filters in ONE launch (the images W and W^T of every trainable 1x1 convolution, once per step): tasks_dev = n
 * lgd_split_task records in device memory -- a: the matrix (element (m, k) at a[m * sm + k * sk], floats), img: its image (lgd_gemm2h_image_bytes(1, M, K),
 * 16-byte aligned), amax: the word holding the float bits of a bound of max |a|, inv: the float that receives the image's inverse scale -- blk0_dev[t] =
 * first workgroup of task t (a workgroup covers 256 of the task's ceil(K/16) * ceil(M/32) * 64 fragment slots), nblocks = their total. */
typedef struct lgd_split_task { unsigned long long a, img, amax, inv; int M, K, sm, sk; } lgd_split_task;
int lgd_gemm2h_split_multi(const void* tasks_dev, const int32_t* blk0_dev, int n, int nblocks, void* stream);
int lgd_gemm2h(const void* image, int image_shared, const float* a_inv, const float* B, const uint32_t* b_amax, long long b_sb, long long b_sk, float* C,
               long long c_sb, long long c_sm, const float* R, long long r_sb, long long r_sm, const float* shift, int relu, uint32_t* relu_bits,
               uint32_t* amax_out, float* splitk_ws, int splits, int nb, int M, int N, int K, void* stream);
/* (splits > 1: split-K for plain products -- no R / shift / relu / mask, C dense -- whose tiles do not fill the chip and whose k-loop is long (the
 *  1024 -> 256 convolutions of res4 at 2 images per GPU: 132 tiles of 64 k-steps): the grid's S row blocks take K / S each and leave partials in
 *  splitk_ws (S * nb * M * N floats), a second launch adds them in fixed order into C and leaves max |C| in amax_out.  splits <= 1: splitk_ws unused.) */

/* ---- K10: the Winograd channel products from f16x2 operands that are split in HBM (csrc/h2.hip; round 5).
 * Replaces, like lgd_gemm3, the arithmetic of every nn.Conv2d(C, C', 3, padding=1) of the path (dynamic_teacher.py:57,61,67-73,145,280;
 * sequential_convs.py:10-12; the head towers, distillator.py:107-109) -- forward M = U V, input gradient dV = U^T dM AND weight gradient
 * dU = dM V^T -- on v_mfma_f32_32x32x16_f16:  x 2^e = h + m (two f16 pieces, |error| <= 2^-23 |x|),  a b ~ ah bh + ah bm + am bh, fp32
 * accumulation; one power-of-two scale per operand and batch (frequency), derived from a guaranteed magnitude bound.
 *   split rows: the format the F(6x6,3x3) transforms write V / dM in (lgd_wino_in_h2, lgd_wino_out_t_h2, lgd_wino_in_t_out_t_h2): 4 bytes per
 *               element like fp32; tile t of a row at (t / 32) * 128 + piece * 64 + (t % 32) * 2 bytes.
 *   image:      the filter operand, [batch][k-step of 16][2 pieces][32-row block][1 KB] (lgd_wino_filter_images_h2), lgd_h2_image_bytes bytes.
 * lgd_h2_fwd:   C[b] (M x N, fp32, C[b * c_sb + m * c_sm + n]) = image[b] (M x K) . B[b], B split rows with k-row stride b_sk and batch stride
 *               b_sb (bytes; extent b_bytes from B), rescaled by a_inv[b] * b_inv[b_inv_per_batch ? b : 0].  amax_out (optional, nb words):
 *               max |C[b]| as float bits (words zeroed by the CALLER).  K % 16 == 0.
 * lgd_h2_dw:    out[b] (M x N, row-major fp32) = sum_t A[b][m][t] B[b][n][t], both split rows (row strides a_rs / b_rs, batch strides a_sb / b_sb,
 *               bytes), T % 32 == 0 (lgd_wino_tiles pads every level to 32); split-K over S workgroups per output tile (lgd_h2_dw_splits) through `partials` (S * nb * M * N floats,
 *               unused for S == 1), reduced in fixed order: bit-reproducible; out == NULL (S > 1)
 *               leaves the partials for lgd_wino_filter_bwd_parts, which adds them while it reads.
 * lgd_h2_amax_maps / lgd_h2_amax_filters: max |act(x)| over the level maps (act = identity, relu(x + bias[c]) or relu(x * scale + shift) with
 *               the (L*N, C, 2) table of lgd_wino_in) / max |w * scale[co]| over K <= 8 filters (word zeroed by the CALLER), as float bits (`accumulate` != 0: the word is NOT zeroed
 *               first -- the maximum over several calls, or a word the caller zeroed): the bounds the transforms derive their scales from.  lgd_h2_link_bound: the bound of |dx| for the fused backward link from the 64
 *               per-frequency maxima of dV (lgd_h2_fwd's amax_out). */
size_t lgd_h2_image_bytes(int nb, int M, int K);
int lgd_h2_fwd(const void* image, const void* B, long long b_sb, long long b_sk, long long b_bytes, float* C, long long c_sb, long long c_sm,
               const float* a_inv, const float* b_inv, int b_inv_per_batch, uint32_t* amax_out, int nb, int M, int N, int K, void* stream);
int lgd_h2_dw_splits(int nb, int M, int N, int T);
int lgd_h2_dw(const void* A, long long a_rs, long long a_sb, long long a_bytes, const float* a_inv, int a_inv_per_batch, const void* B, long long b_rs,
              long long b_sb, long long b_bytes, const float* b_inv, int b_inv_per_batch, float* out, float* partials, int S, int nb, int M, int N, int T,
              void* stream);
/* lgd_h2_pwdw: the weight gradient of a 1x1 convolution, partials[s] (M x N) = sum over split s of the (image, pixel) range of dz[n][m][px] x[n][k][px]
 * (dz (nimg, M, HW), x (nimg, N, HW) fp32 NCHW maps, HW % 4 == 0), both operands scaled by the powers of two their bounds *a_amax / *b_amax prescribe and
 * split into f16 pairs in registers; S = lgd_h2_pwdw_splits(...) partials, added (with the frozen per-row scale) by lgd_sum_batch_scale(partials, scale,
 * S, M, N, dw). */
int lgd_h2_pwdw_splits(int nimg, int M, int N, int HW);
int lgd_h2_pwdw(const float* dz, const float* x, const uint32_t* a_amax, const uint32_t* b_amax, float* partials, int S, int nimg, int M, int N, int HW,
                void* stream);
int lgd_h2_amax_maps(const float* const* x_host, const int32_t* level_hw_host, int L, int N, int C, const float* pre_bias, const float* pre_affine,
                     uint32_t* out_bits, int accumulate, void* stream);
int lgd_h2_amax_filters(const float* const* w_host, const float* const* scale_host, const int32_t* rows_host, int K, int row_elems, uint32_t* out_bits,
                        void* stream);
int lgd_h2_link_bound(const uint32_t* amax64, uint32_t* out_bits, void* stream);
/* *out = max(*out, *words[0], ..., *words[n-1]) on float bits of non-negative floats (n <= 16 device words, host array of pointers): the bounds of
 * several producers' maps combined for one convolution that reads them all (one 64-thread launch) */
int lgd_h2_words_max(const uint32_t* const* words_host, int n, uint32_t* out, void* stream);
/* out[c] = inv[0] * sum over the T tiles of one frequency plane of a split buffer: buf = the plane's row of channel 0, ch_bytes = distance between
 * channels (64 * 4 * T for [C][64][T]), T % 32 == 0.  The bias gradient of a 3x3 convolution on the f16x2 path: plane tile + 3 of dM
 * [ref: the bias of every nn.Conv2d(., ., 3, padding=1) of the path, dynamic_teacher.py:57-73, sequential_convs.py:10-12]. */
int lgd_h2_plane_sums(const void* buf, long long ch_bytes, int C, int T, const float* inv, float* out, void* stream);
/* lgd_wino_out_t_gn writing dM as split rows; *amax_in: the bound lgd_gn_group_bwd_coef(bound_out) leaves */
int lgd_wino_out_t_gn_h2(const float* const* g_host, const float* const* y_host, const float* coef, const int32_t* level_hw_host, int L, int N, int C,
                         void* dM, const uint32_t* amax_in, float* inv_out64, void* stream);
/* The F(6x6,3x3) transforms around them.  *_h2: the frequency buffer WRITTEN is split rows scaled by a power of two derived from *amax_in (float
 * bits of an upper bound of the transform's input magnitude); inv_out receives the inverse scale(s): 1 float (lgd_wino_in_h2: one scale for all
 * frequencies) or 64 (per frequency).  lgd_wino_out_amax / lgd_wino_in_t_amax: lgd_wino_out / lgd_wino_in_t (tile 6) that also leave
 * max |output| in *amax_out (a word zeroed by the CALLER: lgd_amd keeps a pool of zeroed words, one fill per 4096 bounds).  lgd_wino_filter_images_h2: lgd_wino_filter_images writing the f16x2 images. */
int lgd_wino_in_h2(const float* const* x_host, const int32_t* level_hw_host, int L, int N, int C, void* V, const float* pre_bias,
                   const float* pre_affine, void* pre_bits, const uint32_t* amax_in, float* inv_out, void* stream);
int lgd_wino_out_t_h2(const float* const* dy_host, const void* relu_bits, const int32_t* level_hw_host, int L, int N, int C, void* dM,
                      const uint32_t* amax_in, float* inv_out64, void* stream);
int lgd_wino_in_t_out_t_h2(const float* dV, const void* relu_bits, const int32_t* level_hw_host, int L, int N, int C, void* dM,
                           const uint32_t* bound_in, float* inv_out64, void* stream);
int lgd_wino_out_amax(const float* M, const float* bias, const int32_t* level_hw_host, int L, int N, int C, int relu, float* const* y_host,
                      void* relu_bits, uint32_t* amax_out, void* stream);
int lgd_wino_in_t_amax(const float* dV, const int32_t* level_hw_host, int L, int N, int C, float* const* dx_host, const void* pre_bits,
                       uint32_t* amax_out, void* stream);
int lgd_wino_filter_bwd_parts(const float* dU, long long du_plane, long long part_stride, int S, const float* scale, int Co, int Ci, float* dw,
                              void* stream);   /* lgd_wino_filter_bwd (tile 6) over S split-K partials of dU (lgd_h2_dw with out == NULL), added in fixed order */
int lgd_wino_filter_images_h2(const float* w, const float* scale, int Co, int Ci, int row0, int Ct, void* img_fwd, void* img_bwd,
                              const uint32_t* amax_in, float* inv_out64, void* stream);

/* dx = dy where the bit is set, for rows of HW elements with ceil(HW / 32) mask words each */
size_t lgd_relu_rowbits_words(long long rows, int HW);
int lgd_relu_rowbits_bwd(const uint32_t* relu_bits, const float* dy, long long rows, int HW, float* dx, uint32_t* amax_out, void* stream);
/* (amax_out: NULL, or a word the caller zeroed that receives the float bits of max |dx|: the magnitude tag of the masked gradient, from which the
 *  1x1 products that read it derive their f16x2 scales -- lgd_gemm2h, lgd_h2_pwdw) */

/* ------------------------------------------------------------------ FCOS ground-truth assignment
 * [ref: models/customized_detectors/thirdparty_heads/fcos.py:177-284  FCOS.get_ground_truth]
 * One launch for the mini-batch.  shifts: (R,2) location centres (x,y) of all levels concatenated (level l owns
 * level_locs_host[l] consecutive rows); per level: the regression range [size_lo, size_hi] (OBJECT_SIZES_OF_INTEREST,
 * fcos.py:248-252) and the centre-sampling radius in pixels (stride * CENTER_SAMPLING_RADIUS, fcos.py:228; ignored when
 * center_sampling = 0: a location is then a candidate anywhere strictly inside the box, fcos.py:244-246).
 * gt_boxes (T,4) / gt_classes (T,) int64 image-major, img_off (B+1) int32 (device).  Per (image, location): the candidate
 * box of minimal area (first index on ties, fcos.py:254-259) -> class id (num_classes = background), (l,t,r,b) distances to
 * the matched box (box 0 of the image when there is no candidate, as gt_boxes[argmin] does), centerness
 * sqrt(clamp(min(l,r)/max(l,r),0) * clamp(min(t,b)/max(t,b),0)) (fcos.py:268-276).  Images without boxes: background / zeros.
 * Outputs: classes (B,R) int64, deltas (B,R,4) fp32, centerness (B,R) fp32 -- bit-identical to the elementwise definition.
 */
int lgd_fcos_targets(const float* shifts, const int32_t* level_locs_host, const float* size_lo_host, const float* size_hi_host,
                     const float* radius_px_host, int L, int R, const float* gt_boxes, const int64_t* gt_classes,
                     const int32_t* img_off, int B, int T, int num_classes, int center_sampling, int64_t* out_classes,
                     float* out_deltas, float* out_centerness, void* stream);

/* ------------------------------------------------------------------ conv epilogues of the student (FrozenBN folded)
 * [ref: the detectron2 BottleneckBlock of the reference's student: conv -> FrozenBN (a per-channel
 *  affine, folded into the conv weights + bias) [-> += shortcut] -> relu]
 * lgd_bias_act_fwd : out = x + bias[c] (+ residual) [then ReLU]; x, residual, out: (N, C, HW) fp32; bias / residual may be NULL
 *   relu_bits (may be NULL): the linear bitmap [out > 0] -- element e is bit e % 32 of word e / 32, lgd_relu_bits_words(N*C*HW)
 *   uint32 words -- so that the backward reads 1 bit per element instead of the 4-byte output
 * lgd_relu_bits_bwd: dx = dy where the bitmap says the forward output was > 0, else 0 (2 map transfers instead of 3)
 * lgd_relu_mask_bwd: dx = dy where the saved forward output y > 0, else 0 (total elements)
 */
size_t lgd_relu_bits_words(long long total);
/* frozen stem epilogue [d2-memory: BasicStem -- conv1 -> FrozenBN -> relu -> max_pool2d(3, stride 2, padding 1)]:
 * out (N, C, (H-1)/2+1, (W-1)/2+1) = max_pool2d(relu(y + bias[c]), 3, 2, 1) of the conv output y (N, C, H, W) in one pass. Forward only. */
int lgd_stem_bias_relu_maxpool(const float* y, const float* bias, int N, int C, int H, int W, float* out, void* stream);
/* K11 (csrc/stem.hip): the frozen ResNet stem in ONE kernel -- out (N, 64, Hp, Wp) = max_pool2d(relu(conv2d(x, w, stride 2, padding 3) + shift[c]), 3, 2, 1)
 * for x (N, 3, H, W) fp32 and w (64, 3, 7, 7) fp32 (the FrozenBN scale folded in), forward only [d2-memory: detectron2 BasicStem, frozen by
 * MODEL.BACKBONE.FREEZE_AT = 2 in every config of the reference; the student of distillator.py:100-112].  The convolution is an implicit GEMM on
 * v_mfma_f32_32x32x16_f16 from two-piece f16 operands (fp32-class: error vs fp64 ~5e-7 of the output scale); the conv output never reaches memory.
 * lgd_stem7_image: the filter as MFMA fragments (lgd_stem7_image_bytes() bytes, 16-byte aligned) scaled by the power of two that the bound *w_amax
 * (float bits of max |w|) prescribes, inverse scale in *w_inv -- once per (frozen) filter.  lgd_stem7_conv_pool: *x_amax = float bits of a bound of
 * max |x| (lgd_h2_amax_maps); Ho = (H - 1) / 2 + 1, Hp = (Ho - 1) / 2 + 1 (W alike); amax_out: NULL, or a word the caller zeroed that receives the
 * float bits of max |out| (the tag res2's first 1x1 products take their f16x2 scale from). */
size_t lgd_stem7_image_bytes(void);
int lgd_stem7_image(const float* w, const uint32_t* w_amax, void* image, float* w_inv, void* stream);
int lgd_stem7_conv_pool(const float* x, const void* image, const float* w_inv, const uint32_t* x_amax, const float* shift, int N, int H, int W, float* out,
                        uint32_t* amax_out, void* stream);
int lgd_bias_act_fwd(const float* x, const float* bias, const float* residual, int N, int C, int HW, int relu, float* out,
                     uint32_t* relu_bits, void* stream);
/* weight gradient of a pointwise convolution from per-image partial products: out[o][i] = scale[o] * sum_n part[n][o][i]
 * (scale may be NULL); part (N, Co, Ci), out (Co, Ci). */
int lgd_sum_batch_scale(const float* part, const float* scale, int N, int Co, int Ci, float* out, void* stream);
/* out_t[r][c] = w_t[r][c] * scale_t[r] for a table of n tensors in one launch (the w * scale filter folds of every trainable 1x1
 * convolution in front of a FrozenBN, once per step): tasks_dev = n lgd_rows_task records in device memory, blk0_dev[t] = first
 * workgroup of task t (a workgroup covers 1024 elements), nblocks = their total.  amax_out: NULL, or n words the caller zeroed: the float bits
 * of max |out_t| per task (the f16x2 scale of the filter's image, lgd_gemm2h_split). */
typedef struct lgd_rows_task { unsigned long long w, scale, out; int rows, cols; } lgd_rows_task;
int lgd_scale_rows_multi(const void* tasks_dev, const int32_t* blk0_dev, int n, int nblocks, uint32_t* amax_out, void* stream);
int lgd_relu_bits_bwd(const uint32_t* relu_bits, const float* dy, long long total, float* dx, void* stream);
int lgd_relu_mask_bwd(const float* y, const float* dy, long long total, float* dx, void* stream);
/* y = x[..., ::2, ::2] of `planes` (image, channel) planes of H x W (the input of a 1x1 / stride 2 convolution: detectron2 BottleneckBlock with
 * STRIDE_IN_1X1, projection shortcuts) as ceil(H/2) x ceil(W/2) planes, and its adjoint dx = g at the even (row, column) positions, 0 elsewhere:
 * one pass each. */
int lgd_subsample2_fwd(const float* x, long long planes, int H, int W, float* y, void* stream);
int lgd_subsample2_bwd(const float* g, long long planes, int H, int W, float* dx, void* stream);

/* ------------------------------------------------------------------ anchor <-> ground-truth matching of the student loss
 * [ref: distillator.py:96-112 -> student.losses on student AND teacher features; detectron2 RetinaNet.label_anchors /
 *  Matcher semantics: IoU thresholds [lo, hi] -> labels {0, -1, 1}, allow_low_quality_matches,
 *  background -> num_classes, ignore -> -1, every anchor also gets the box of its arg-max ground truth]
 * anchors (R,4) xyxy; gt_boxes (T,4) / gt_classes (T, int64) concatenated image-major with img_off (B+1, device int32);
 * best_ws: T uint32 scratch; labels (B,R) int64; matched_boxes (B,R,4) fp32.  Two launches for the whole mini-batch, the
 * IoU matrix is never materialised; integer results are bit-identical to the elementwise fp32 definition.
 */
int lgd_anchor_match(const float* anchors, int R, const float* gt_boxes, const int64_t* gt_classes, const int32_t* img_off,
                     int B, int T, float iou_lo, float iou_hi, int num_classes, int allow_low_quality, uint32_t* best_ws,
                     int64_t* labels, float* matched_boxes, void* stream);

/* ------------------------------------------------------------------ box-regression loss on the head's raw NCHW deltas
 * [ref: distillator.py:107-112 -> student.losses -> detectron2 RetinaNet.losses: smooth_l1(pred_deltas[pos],
 *  Box2BoxTransform(weights).get_deltas(anchors, matched_gt)[pos], beta, reduction="sum")]
 * deltas_host: L device pointers to (N, A*4, H_l, W_l); labels_host: the int32 label planes (N, A, H_l, W_l) of the focal
 * loss (positives: 0 <= label < K); anchors (R,4) ordered (level, y, x, a); matched_boxes (N,R,4) from lgd_anchor_match;
 * ws: lgd_box_reg_ws_doubles(...) doubles; loss: 1 fp32.  backward writes dense gradients (zeros off the positives).
 */
size_t lgd_box_reg_ws_doubles(const int32_t* level_hw_host, int L, int N, int A);
int lgd_box_reg_loss_fwd(const float* const* deltas_host, const int32_t* const* labels_host, const int32_t* level_hw_host, int L,
                         int N, int A, int K, const float* anchors, const float* matched_boxes, int R, float beta,
                         const float* weights4_host, double* ws, float* loss, void* stream);
int lgd_box_reg_loss_bwd(const float* const* deltas_host, const int32_t* const* labels_host, const int32_t* level_hw_host, int L,
                         int N, int A, int K, const float* anchors, const float* matched_boxes, int R, float beta,
                         const float* weights4_host, const float* grad_loss, float* const* grad_deltas_host, void* stream);

/* ------------------------------------------------------------------ modulated deformable 3x3 convolution (DCNv2, config 5)
 * [ref: configs/Distillation/RetinaNet/retinanet_R_101_dcnv2_*.yaml:7-8 -> detectron2 ModulatedDeformConv]
 * lgd_dcn_im2col: col[n][c*9+k][y*Wo+x] = mask[n,k,y,x] * bilinear(x[n,c], y*s - p + ky*d + off[n,2k,y,x], x*s - p + kx*d + off[n,2k+1,y,x])
 *   (zero outside the map; mask may be NULL = 1); the convolution is then W (O x C*9) @ col, a library GEMM issued by the host.
 * lgd_dcn_col2im: from d col (same layout) the gradients dx (N,C,H,W; zeroed here -- dx, d offset, d mask and ws carved in this order out
 *   of one allocation, gaps under 16 bytes, are zeroed by ONE fill instead of four; atomic scatter), d offset (N,18,Ho,Wo) and
 *   d mask (N,9,Ho,Wo; ignored when mask is NULL).  Ho = (H + 2p - 2d - 1)/s + 1, likewise Wo.
 */
int lgd_dcn_im2col(const float* x, const float* offset, const float* mask, int N, int C, int H, int W, int stride, int pad,
                   int dilation, float* col, void* stream);
size_t lgd_dcn_ws_bytes(int N, int H, int W);
int lgd_dcn_col2im(const float* x, const float* offset, const float* mask, const float* dcol, int N, int C, int H, int W,
                   int stride, int pad, int dilation, float* dx, float* doffset, float* dmask,
                   void* ws /* lgd_dcn_ws_bytes(N, H, W) bytes: dx by gather through per-cell contribution lists; NULL: dx by atomic scatter */,
                   void* stream);
/* packed forms: om / dom (N, 27, Ho, Wo) = the offset convolution's own output and its gradient -- channels 0..17 the offsets, 18..26 the
 * mask LOGITS (detectron2 DeformBottleneckBlock: chunk(3), offset = cat(o1, o2), mask = sigmoid(m)); the kernels read om in place, apply
 * the sigmoid and write d logit = d mask * m (1 - m): no cat / sigmoid passes in either direction. */
int lgd_dcn_im2col_packed(const float* x, const float* om, int N, int C, int H, int W, int stride, int pad, int dilation, float* col,
                          void* stream);
int lgd_dcn_col2im_packed(const float* x, const float* om, const float* dcol, int N, int C, int H, int W, int stride, int pad,
                          int dilation, float* dx, float* dom, void* ws, void* stream);

/* ------------------------------------------------------------------ gradient clipping + SGD of both optimizers, one launch
 * [ref: train.py:200-204 stu_optimizer.step() / tea_optimizer.step(); utils/build.py:494-529 torch.optim.SGD(momentum, weight
 *  decay) wrapped by detectron2's per-parameter gradient clipping, CLIP_TYPE "value"]
 * For every tensor of the table (device memory, n_tensors entries):
 *     g <- clamp(g, -clip_value, clip_value);  buf <- mu * buf + (g + wd * p);  p <- p - lr * buf
 * (torch.optim.SGD's update with dampening 0, nesterov off; a zero-initialised buf reproduces its first step exactly).
 * blk_off (device, n_tensors + 1 int32): workgroup b works on tensor i with blk_off[i] <= b < blk_off[i+1], on elements
 * [(b - blk_off[i]) * lgd_sgd_chunk_elems(), ...); n_blocks = blk_off[n_tensors].  clip_value = +inf: no clipping.
 */
typedef struct lgd_sgd_tensor {
    float* p;          /* parameter */
    float* g;          /* its gradient (the clipped value is written back) */
    float* m;          /* momentum buffer */
    int64_t n;         /* elements */
    float lr, wd, mu;  /* learning rate, weight decay, momentum of the optimizer that owns the tensor */
    int32_t reserved;
} lgd_sgd_tensor;
int lgd_sgd_chunk_elems(void);
int lgd_sgd_clip_step(const lgd_sgd_tensor* table, const int32_t* blk_off, int n_tensors, int n_blocks, float clip_value,
                      void* stream);

/* ------------------------------------------------------------------ per-kernel timing (bench.py)
 * When enabled every kernel launch of this library is bracketed by a HIP event pair recorded on
 * the launch stream.  lgd_timing_collect waits for the recorded events, sums the elapsed time per
 * kernel name and clears the records: names = NUL-separated kernel names, total_ms/launches per
 * name; returns the number of names written. */
int lgd_timing_enable(int on);
int lgd_timing_collect(char* names, size_t names_len, double* total_ms, int32_t* launches, int max_entries);
/* same, plus the shortest / longest launch of every kernel name (ms) */
int lgd_timing_collect_ex(char* names, size_t names_len, double* total_ms, double* min_ms, double* max_ms, int32_t* launches,
                          int max_entries);
/* Stall diagnosis (round 6): the recorded launches whose end event has not completed yet, oldest first, one text line each --
 * "RUNNING <stream> <kernel>" (start event done) or "QUEUED <stream> <kernel>"; never blocks; returns their number.  There is no
 * counterpart in the reference (it has no kernels of its own: SURVEY.md section 2a); the step it watches is train.py:182-215. */
int lgd_timing_pending(char* out, size_t out_len);

#ifdef __cplusplus
}
#endif
#endif /* LGD_HIP_H */
