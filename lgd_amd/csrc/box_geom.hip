// Box geometry: fp32-exact inside-box rectangles and row bands for every (FPN level, box).
// Replaces the dense float masks of the reference
//   [ref: models/customized_detectors/dynamic_teacher/utils.py:53-89 get_inside_gt_mask]
// One workgroup per (level, image); threads own boxes, then sort the row breakpoints.
#include "common.h"

#pragma clang fp contract(off)  // the inclusion predicate must round exactly like the reference's separate torch ops

namespace lgd {

static thread_local hipError_t g_last_err = hipSuccess;
void set_last_error(hipError_t e) { g_last_err = e; }

struct PrepArgs {
    const float* boxes;
    const int32_t* img_off;
    int32_t* geom;
    int L, B, T, max_n, img_h, img_w;
    int H[LGD_MAX_LEVELS], W[LGD_MAX_LEVELS];
};

// [lo, hi] = the integer coordinates p in [0, n) with |c - p| / s <= 0.5 evaluated in fp32 exactly as torch does
// (abs, IEEE divide, compare).  The true-set is an interval because every operation is monotone in |c - p|; an
// empty set returns hi < lo.  One WAVE evaluates one box axis: lane j tests coordinates j, j+64, ... and the
// interval ends come from the ballots (no serial loop over the pixels).
__device__ __forceinline__ void axis_interval(float a, float b, float ratio, int n, int lane, int& lo, int& hi) {
    // plain operators under the file-scope contract(off): the header intrinsics (__fmul_rn, ...) are inline functions
    // compiled with the default fp-contract=fast and were fused into v_fma after inlining (a*r + b1 in ONE rounding)
    const float a1 = a * ratio;        // box_tensor[:, [0,2]] * r_w  (utils.py:69)
    const float b1 = b * ratio;
    const float c = (a1 + b1) * 0.5f;  // (x1 + x2) * 0.5             (utils.py:73)
    const float s = b1 - a1;           // x2 - x1                     (utils.py:77)
    lo = n;
    hi = -1;
    for (int p0 = 0; p0 < n; p0 += 64) {
        const int p = p0 + lane;
        const float d = fabsf(c - (float)p) / s;  // utils.py:87 ('/' = correctly rounded division)
        // false for NaN (0/0) and +inf (x/0): zero-extent boxes are empty
        const unsigned long long m = __ballot(p < n && d <= 0.5f);
        if (m) {
            lo = min(lo, p0 + (int)__builtin_ctzll(m));
            hi = max(hi, p0 + 63 - (int)__builtin_clzll(m));
        }
    }
}

__global__ __launch_bounds__(256) void box_prep_kernel(PrepArgs a) {
    extern __shared__ __attribute__((aligned(16))) int smem[];  // 3 x [2*max_n+2]: keys | sorted | isfirst
    const int l = blockIdx.x / a.B, b = blockIdx.x % a.B;
    const int H = a.H[l], W = a.W[l];
    const int t0 = a.img_off[b], n = a.img_off[b + 1] - t0;
    const int maxbp = geom_maxbp(a.max_n);
    int* keys = smem;
    int* sorted = smem + maxbp;
    int* isfirst = smem + 2 * maxbp;
    // Python computes dst/src in double, the tensor multiply then uses it as an fp32 scalar (utils.py:66-70).
    const float r_w = (float)((double)W / (double)a.img_w);
    const float r_h = (float)((double)H / (double)a.img_h);
    int32_t* rects = a.geom + geom_rects_off() + ((size_t)l * a.B + b) * a.max_n * 4;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = wave; i < n; i += 4) {  // one wave per box
        const float4 bx = reinterpret_cast<const float4*>(a.boxes)[t0 + i];
        int x0, x1, y0, y1;
        axis_interval(bx.x, bx.z, r_w, W, lane, x0, x1);
        axis_interval(bx.y, bx.w, r_h, H, lane, y0, y1);
        if (x1 < x0 || y1 < y0) { x0 = 0; x1 = -1; y0 = 0; y1 = -1; }
        if (lane == 0) {
            reinterpret_cast<int4*>(rects)[i] = make_int4(x0, x1, y0, y1);
            // row breakpoints: a band starts where a box starts and right after it ends
            keys[2 * i] = (y1 >= y0) ? y0 : 0;
            keys[2 * i + 1] = (y1 >= y0) ? y1 + 1 : H;
        }
    }
    if (threadIdx.x == 0) { keys[2 * n] = 0; keys[2 * n + 1] = H; }
    __syncthreads();
    const int m = 2 * n + 2;
    // de-duplicating rank sort: a key survives iff it is the first occurrence of its value;
    // its rank is the number of surviving smaller keys.
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const int v = keys[i];
        int first = 1;
        for (int j = 0; j < i; ++j) first &= (keys[j] != v);
        isfirst[i] = first;
        sorted[i] = -1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        if (!isfirst[i]) continue;
        const int v = keys[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += (isfirst[j] && keys[j] < v) ? 1 : 0;
        sorted[rank] = v;
    }
    __syncthreads();
    int32_t* bands = a.geom + geom_bands_off(a.L, a.B, a.max_n) + ((size_t)l * a.B + b) * maxbp;
    int cnt = 0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const int v = sorted[i];
        if (v >= 0) bands[i] = v;
    }
    if (threadIdx.x == 0) {
        for (int i = 0; i < m; ++i) cnt += sorted[i] >= 0 ? 1 : 0;  // distinct values occupy sorted[0..cnt)
        a.geom[geom_nbp_off(a.L, a.B, a.max_n) + (size_t)l * a.B + b] = cnt;
    }
}

}  // namespace lgd

extern "C" {

int lgd_abi_version(void) { return 26; }
const char* lgd_arch(void) { return "gfx950"; }
const char* lgd_last_error(void) {
    return lgd::g_last_err == hipSuccess ? "" : hipGetErrorString(lgd::g_last_err);
}

size_t lgd_geom_rects_off(int, int, int, int) { return lgd::geom_rects_off(); }
size_t lgd_geom_nbp_off(int L, int B, int, int max_n) { return lgd::geom_nbp_off(L, B, max_n); }
size_t lgd_geom_bands_off(int L, int B, int, int max_n) { return lgd::geom_bands_off(L, B, max_n); }
size_t lgd_geom_ints(int L, int B, int, int max_n) {
    return lgd::geom_bands_off(L, B, max_n) + (size_t)L * B * lgd::geom_maxbp(max_n);
}

int lgd_box_prep(const float* boxes, const int32_t* img_off, int B, int T, int max_n, int img_h, int img_w,
                 const int32_t* level_hw_host, int L, int32_t* geom, void* stream) {
    if (!boxes || !img_off || !geom || !level_hw_host || L < 1 || L > LGD_MAX_LEVELS || B < 1 || T < 0 || max_n < 0)
        return LGD_EINVAL;
    lgd::PrepArgs a;
    a.boxes = boxes; a.img_off = img_off; a.geom = geom;
    a.L = L; a.B = B; a.T = T; a.max_n = max_n; a.img_h = img_h; a.img_w = img_w;
    for (int l = 0; l < L; ++l) { a.H[l] = level_hw_host[2 * l]; a.W[l] = level_hw_host[2 * l + 1]; }
    const size_t smem = 3 * (size_t)lgd::geom_maxbp(max_n) * sizeof(int);
    if (smem > 160 * 1024) return LGD_EINVAL;
    LGD_LAUNCH("box_prep_kernel", lgd::box_prep_kernel, dim3(L * B), dim3(256), smem, (hipStream_t)stream, a);
    return lgd::check_launch();
}

}  // extern "C"
