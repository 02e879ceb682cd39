// One C-ABI call per 3x3 convolution: filter transform -> input transform -> per-frequency channel GEMMs (rocBLAS fp32 MFMA, issued by
// this library on the caller's stream with the solution index the host passes from its tuning table) -> output transform, and the
// matching backward (dy expansion -> dV / dU GEMMs -> adjoint input transform or the fused link -> filter-gradient transform -> bias
// gradient).  [ref: every nn.Conv2d(., ., 3, padding=1) of the path: dynamic_teacher.py:57,61,67-73, sequential_convs.py:10-12,
// distillator.py:107-109 -> retinanet.py:36-43]
// Why: at 2 images per GPU (the per-rank workload of the reference's 8-GPU recipe) a convolution's kernels run 10-90 us each and the
// host needed 6-12 Python-level calls (ctypes launches, torch.bmm, allocations) per convolution and direction; here it is one call.
// The library never allocates device memory on the data path: every buffer, including the GEMM operands, is the caller's.  rocBLAS keeps
// a handle (created once, lgd_blas_init) whose own device workspace is allocated at creation.
#include <mutex>

#include <rocblas/rocblas.h>

#include "winograd.h"

namespace lgd {

static rocblas_handle g_blas = nullptr;
static std::mutex g_blas_mu;

static int blas_handle(rocblas_handle* out) {
    std::lock_guard<std::mutex> lk(g_blas_mu);
    if (!g_blas) {
        if (rocblas_create_handle(&g_blas) != rocblas_status_success) { g_blas = nullptr; return LGD_ELAUNCH; }
        rocblas_set_atomics_mode(g_blas, rocblas_atomics_not_allowed);   // fixed summation order: bitwise reproducible products
        rocblas_set_pointer_mode(g_blas, rocblas_pointer_mode_host);
    }
    *out = g_blas;
    return LGD_OK;
}

// the three per-frequency products of a Winograd convolution on the [C][nf][T] buffers (row-major views; rocBLAS is column-major, so
// the operands are swapped):  kind 0  M[f]  (Ct x T)  = U[f]  (Ct x Ci) . V[f] (Ci x T)
//                             kind 1  dV[f] (Ci x T)  = Ut[f] (Ci x Ct) . dM[f] (Ct x T)
//                             kind 2  dU[f] (Ct x Ci) = dM[f] (Ct x T)  . V[f]^T
// solution: the rocBLAS solution index the host's tuning table holds for this shape (0: the library's own choice); an index the
// loaded rocBLAS does not know falls back to its own choice.
static int wino_gemm(const char* name, int kind, const float* A, const float* B, float* C, int Ct, int Ci, long long T, int nf, int solution,
                     hipStream_t st) {
    rocblas_handle h;
    if (blas_handle(&h) != LGD_OK) return LGD_ELAUNCH;
    if (rocblas_set_stream(h, st) != rocblas_status_success) return LGD_ELAUNCH;
    const float one = 1.f, zero = 0.f;
    const long long ldt = (long long)nf * T;
    rocblas_operation ta = rocblas_operation_none, tb = rocblas_operation_none;
    rocblas_int m, n, k, lda, ldb, ldc;
    rocblas_stride sa, sb, sc;
    const float *a, *b;
    if (ldt > 0x7fffffffLL || T > 0x7fffffffLL) return LGD_EINVAL;
    if (kind == 0) {        // C^T (T x Ct) = V^T (T x Ci) . U^T (Ci x Ct)
        m = (rocblas_int)T; n = Ct; k = Ci; a = B; lda = (rocblas_int)ldt; sa = T; b = A; ldb = Ci; sb = (rocblas_stride)Ct * Ci; ldc = (rocblas_int)ldt; sc = T;
    } else if (kind == 1) { // dV^T (T x Ci) = dM^T (T x Ct) . Ut^T (Ct x Ci)
        m = (rocblas_int)T; n = Ci; k = Ct; a = B; lda = (rocblas_int)ldt; sa = T; b = A; ldb = Ct; sb = (rocblas_stride)Ci * Ct; ldc = (rocblas_int)ldt; sc = T;
    } else {                // dU^T (Ci x Ct) = V (as column-major T x Ci, transposed) . dM^T (T x Ct)
        ta = rocblas_operation_transpose;
        m = Ci; n = Ct; k = (rocblas_int)T; a = B; lda = (rocblas_int)ldt; sa = T; b = A; ldb = (rocblas_int)ldt; sb = T; ldc = Ci; sc = (rocblas_stride)Ct * Ci;
    }
    KTimer timer(name, st);
    rocblas_status rs = rocblas_status_invalid_value;
    if (solution != 0)
        rs = rocblas_gemm_strided_batched_ex(h, ta, tb, m, n, k, &one, a, rocblas_datatype_f32_r, lda, sa, b, rocblas_datatype_f32_r, ldb, sb, &zero,
                                             C, rocblas_datatype_f32_r, ldc, sc, C, rocblas_datatype_f32_r, ldc, sc, nf, rocblas_datatype_f32_r,
                                             rocblas_gemm_algo_solution_index, solution, rocblas_gemm_flags_none);
    if (rs != rocblas_status_success)
        rs = rocblas_gemm_strided_batched_ex(h, ta, tb, m, n, k, &one, a, rocblas_datatype_f32_r, lda, sa, b, rocblas_datatype_f32_r, ldb, sb, &zero,
                                             C, rocblas_datatype_f32_r, ldc, sc, C, rocblas_datatype_f32_r, ldc, sc, nf, rocblas_datatype_f32_r,
                                             rocblas_gemm_algo_standard, 0, rocblas_gemm_flags_none);
    return rs == rocblas_status_success ? LGD_OK : LGD_ELAUNCH;
}

// db[c] = sum_t dM[c][tile + 3][t]: A's row of the interpolation point 1 is all ones, so that frequency of dM = A g A^T is the tile's
// gradient sum and its row sum the bias gradient.  One wave per channel, fixed order.
__global__ __launch_bounds__(256) void wino_bias_grad_kernel(const float* dM, long long cs, long long T, int f, int C, float* db) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    const float* p = dM + (size_t)c * cs + (size_t)f * T;
    float s = 0.f;
    for (long long t = lane * 4; t < T; t += 256) {   // T is a multiple of 16
        const float4 v = *reinterpret_cast<const float4*>(p + t);
        s += (v.x + v.y) + (v.z + v.w);
    }
    s = wave_sum(s);
    if (lane == 0) db[c] = s;
}

}  // namespace lgd

extern "C" {

int lgd_blas_init(void) {
    rocblas_handle h;
    return lgd::blas_handle(&h);
}

int lgd_blas_version(char* buf, size_t len) {
    if (!buf || len < 2) return LGD_EINVAL;
    size_t need = 0;
    if (rocblas_get_version_string_size(&need) != rocblas_status_success || need > len) return LGD_EINVAL;
    return rocblas_get_version_string(buf, len) == rocblas_status_success ? LGD_OK : LGD_ELAUNCH;
}

int lgd_wino_gemm(int kind, const float* A, const float* B, float* C, int Ct, int Ci, long long T, int tile, int solution, void* stream) {
    if (!A || !B || !C || kind < 0 || kind > 2 || Ct < 1 || Ci < 1 || T < 1 || (tile != 4 && tile != 6)) return LGD_EINVAL;
    static const char* names[3] = {"wino_gemm_fwd", "wino_gemm_dx", "wino_gemm_dw"};
    return lgd::wino_gemm(names[kind], kind, A, B, C, Ct, Ci, T, (tile + 2) * (tile + 2), solution, (hipStream_t)stream);
}

int lgd_conv3x3_fwd(const lgd_conv3x3_fwd_args* a, void* stream) {
    if (!a || a->K < 1 || a->K > LGD_MAX_FILTERS || a->L < 1 || a->L > LGD_MAX_LEVELS || !a->U || !a->Ut || !a->V || !a->M) return LGD_EINVAL;
    const int tile = a->tile, nf = (tile + 2) * (tile + 2), K = a->K, L = a->L;
    int Ct = 0;
    for (int k = 0; k < K; ++k) { if (a->Co[k] < 1 || !a->w[k]) return LGD_EINVAL; Ct += a->Co[k]; }
    const long long T = (long long)lgd_wino_tiles(a->level_hw, L, a->N, tile);
    if (T <= 0) return LGD_EINVAL;
    int rc, c0 = 0;
    for (int k = 0; k < K; ++k) {   // U / U^T of the K filters stacked along C_out
        rc = lgd_wino_filter_fwd(a->w[k], a->scale[k], a->Co[k], a->Ci, tile, a->U + (size_t)c0 * a->Ci, (long long)Ct * a->Ci, a->Ut + c0, Ct,
                                 (long long)a->Ci * Ct, stream);
        if (rc != LGD_OK) return rc;
        c0 += a->Co[k];
    }
    rc = lgd_wino_in(a->x, a->level_hw, L, a->N, a->Ci, tile, a->V, a->pre_bias, nullptr, a->pre_bits, stream);
    if (rc != LGD_OK) return rc;
    rc = lgd::wino_gemm("wino_gemm_fwd", 0, a->U, a->V, a->M, Ct, a->Ci, T, nf, a->sol_fwd, (hipStream_t)stream);
    if (rc != LGD_OK) return rc;
    const size_t mask_bytes = lgd_wino_mask_bytes(tile);
    c0 = 0;
    for (int k = 0; k < K; ++k) {   // one output transform per filter: its channels are a contiguous slab of M ([C][nf][T])
        rc = lgd_wino_out(a->M + (size_t)c0 * nf * T, a->bias[k], a->level_hw, L, a->N, a->Co[k], tile, a->relu, a->y + (size_t)k * L,
                          a->relu_bits ? (char*)a->relu_bits + (size_t)c0 * T * mask_bytes : nullptr, stream);
        if (rc != LGD_OK) return rc;
        c0 += a->Co[k];
    }
    return LGD_OK;
}

int lgd_conv3x3_bwd(const lgd_conv3x3_bwd_args* a, void* stream) {
    if (!a || a->K < 1 || a->K > LGD_MAX_FILTERS || a->L < 1 || a->L > LGD_MAX_LEVELS || !a->dM || !a->Ut) return LGD_EINVAL;
    const int tile = a->tile, nf = (tile + 2) * (tile + 2), K = a->K, L = a->L;
    int Ct = 0;
    for (int k = 0; k < K; ++k) { if (a->Co[k] < 1) return LGD_EINVAL; Ct += a->Co[k]; }
    const long long T = (long long)lgd_wino_tiles(a->level_hw, L, a->N, tile);
    if (T <= 0) return LGD_EINVAL;
    const size_t mask_bytes = lgd_wino_mask_bytes(tile);
    hipStream_t st = (hipStream_t)stream;
    int rc, c0 = 0;
    if (!a->dM_ready) {   // dy is expanded ONCE: dM = A (dy . mask) A^T, per filter into its slab
        for (int k = 0; k < K; ++k) {
            rc = lgd_wino_out_t(a->dy + (size_t)k * L, a->relu_bits ? (const char*)a->relu_bits + (size_t)c0 * T * mask_bytes : nullptr, a->level_hw, L,
                                a->N, a->Co[k], tile, a->dM + (size_t)c0 * nf * T, stream);
            if (rc != LGD_OK) return rc;
            c0 += a->Co[k];
        }
    }
    bool need_w = false, need_b = false;
    for (int k = 0; k < K; ++k) { need_w |= a->dw[k] != nullptr; need_b |= a->db[k] != nullptr; }
    if (need_w) {   // weight gradient first: dM is overwritten by nothing below, V is read once
        if (!a->V || !a->dU) return LGD_EINVAL;
        rc = lgd::wino_gemm("wino_gemm_dw", 2, a->dM, a->V, a->dU, Ct, a->Ci, T, nf, a->sol_dw, st);
        if (rc != LGD_OK) return rc;
        c0 = 0;
        for (int k = 0; k < K; ++k) {
            if (a->dw[k]) {
                rc = lgd_wino_filter_bwd(a->dU + (size_t)c0 * a->Ci, (long long)Ct * a->Ci, a->scale[k], a->Co[k], a->Ci, tile, a->dw[k], stream);
                if (rc != LGD_OK) return rc;
            }
            c0 += a->Co[k];
        }
    }
    if (need_b) {
        c0 = 0;
        for (int k = 0; k < K; ++k) {
            if (a->db[k]) {
                LGD_LAUNCH("wino_bias_grad_kernel", lgd::wino_bias_grad_kernel, dim3((a->Co[k] + 3) / 4), dim3(256), 0, st,
                           a->dM + (size_t)c0 * nf * T, (long long)nf * T, T, tile + 3, a->Co[k], a->db[k]);
            }
            c0 += a->Co[k];
        }
        rc = lgd::check_launch();
        if (rc != LGD_OK) return rc;
    }
    if (a->dV) {   // input gradient in the frequency domain, brought back by the adjoint input transform -- or handed straight on
        rc = lgd::wino_gemm("wino_gemm_dx", 1, a->Ut, a->dM, a->dV, Ct, a->Ci, T, nf, a->sol_dx, st);
        if (rc != LGD_OK) return rc;
        if (a->dM_prev) rc = lgd_wino_in_t_out_t(a->dV, a->prev_bits, a->level_hw, L, a->N, a->Ci, tile, a->dM_prev, stream);
        else rc = lgd_wino_in_t(a->dV, a->level_hw, L, a->N, a->Ci, tile, a->dx, a->pre_bits, stream);
        if (rc != LGD_OK) return rc;
    }
    return LGD_OK;
}

}  // extern "C"
