// Modulated deformable 3x3 convolution (DCNv2) of BASELINE config 5 (RetinaNet R-101-DCNv2)  
//   [ref: configs/Distillation/RetinaNet/retinanet_R_101_dcnv2_*.yaml:7-8 DEFORM_ON_PER_STAGE / DEFORM_MODULATED ->
//    detectron2 ModulatedDeformConv]:
//   out[n,o,y,x] = sum_{c,k} W[o,c,k] * mask[n,k,y,x] * bilinear(in[n,c], y*s - p + ky*d + dy_k, x*s - p + kx*d + dx_k)
// with zero padding outside the input and offsets stored as (dy, dx) channel pairs per tap k = ky*3 + kx.
// The per-tap grid_sample formulation costs 9 sampling passes + their autograd per conv (8 ms per conv at res3, 250 ms of
// the 300 ms step).  Here the sampling is one gather kernel that writes the column matrix col (N, C*9, Ho*Wo) once; the
// channel contraction is a library GEMM (W (O x C*9) @ col), and the backward is one kernel that turns d col into dx
// (atomic scatter of the bilinear weights), d offset and d mask (register sums over the channels, no atomics).
#include "common.h"

namespace lgd {

constexpr int kDcnSlots = 8;   // list slots per (input cell, tap): a regular sampling grid fills 4

struct DcnArgs {
    const float* x;       // (N, C, H, W)
    const float* offset;  // (N, 18, Ho, Wo)
    const float* mask;    // (N, 9, Ho, Wo) or null (unmodulated: 1)
    float* col;           // (N, C*9, Ho*Wo)
    const float* dcol;    // backward input, same layout as col
    float* dx;            // (N, C, H, W), zero-initialised by the caller
    float* doffset;       // (N, 18, Ho, Wo)
    float* dmask;         // (N, 9, Ho, Wo) or null
    long long off_bs, mask_bs, doff_bs, dmask_bs;   // batch strides (floats): 18 / 9 HoWo for separate tensors, 27 HoWo when offset and mask
                                                     //   are channels 0..17 / 18..26 of the offset convolution's own output (packed form)
    int sig;              // packed form: the mask channels hold LOGITS (mask = sigmoid, d mask -> d logit in the backward epilogue)
    int N, C, H, W, Ho, Wo, stride, pad, dil;
    int cchunk;           // channels per blockIdx.y slice of the backward kernel (C if not split)
    int* cnt;             // gather path: [N][9][H*W] contributions of tap k that land in an input cell
    int2* ent;            //              [N][9][kDcnSlots][H*W] (source pixel, weight bits) of the first kDcnSlots of them
};

__device__ __forceinline__ float dcn_mask(const DcnArgs& a, int n, int k, int pix) {
    if (!a.mask) return 1.f;
    const float v = a.mask[(size_t)n * a.mask_bs + (size_t)k * (a.Ho * a.Wo) + pix];
    return a.sig ? 1.f / (1.f + expf(-v)) : v;
}

struct Bilin { int y0, x0; float wy1, wx1; bool in; };
__device__ __forceinline__ Bilin bilin_setup(float py, float px, int H, int W) {
    Bilin b;
    b.in = py > -1.f && px > -1.f && py < (float)H && px < (float)W;
    const float fy = floorf(py), fx = floorf(px);
    b.y0 = (int)fy; b.x0 = (int)fx;
    b.wy1 = py - fy; b.wx1 = px - fx;
    return b;
}
__device__ __forceinline__ float at(const float* p, int y, int x, int H, int W) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? p[(size_t)y * W + x] : 0.f;
}

// thread per (n, tap, pixel) and channel slice (blockIdx.y): the sampling geometry (offsets, bilinear weights, border tests, mask)
// is computed once and serves every channel of the slice -- per channel four gathered loads, one fma chain, one coalesced store
// (thread per (n, c, pixel) recomputed the geometry of all 9 taps for every channel: 56 us per launch at config 5)
__global__ __launch_bounds__(256) void dcn_im2col_kernel(DcnArgs a) {
    const int HoWo = a.Ho * a.Wo;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.N * 9 * HoWo) return;
    const int pix = (int)(i % HoWo), k = (int)((i / HoWo) % 9), n = (int)(i / ((long long)HoWo * 9));
    const int ho = pix / a.Wo, wo = pix % a.Wo, ky = k / 3, kx = k % 3;
    const float* off = a.offset + (size_t)n * a.off_bs + pix;
    const float m = dcn_mask(a, n, k, pix);
    const float py = (float)(ho * a.stride - a.pad + ky * a.dil) + off[(size_t)(2 * k) * HoWo];
    const float pxx = (float)(wo * a.stride - a.pad + kx * a.dil) + off[(size_t)(2 * k + 1) * HoWo];
    const Bilin b = bilin_setup(py, pxx, a.H, a.W);
    const float wy0 = 1.f - b.wy1, wx0 = 1.f - b.wx1;
    const bool y0ok = b.in && b.y0 >= 0 && b.y0 < a.H, y1ok = b.in && b.y0 + 1 >= 0 && b.y0 + 1 < a.H;
    const bool x0ok = b.x0 >= 0 && b.x0 < a.W, x1ok = b.x0 + 1 >= 0 && b.x0 + 1 < a.W;
    const bool ok00 = y0ok && x0ok, ok01 = y0ok && x1ok, ok10 = y1ok && x0ok, ok11 = y1ok && x1ok;
    const long long o00 = (long long)b.y0 * a.W + b.x0;          // may be negative where the sample hangs over the border
    const int c0 = blockIdx.y * a.cchunk, c1 = min(a.C, c0 + a.cchunk);
    const size_t HW = (size_t)a.H * a.W;
    const float* p = a.x + ((size_t)n * a.C + c0) * HW;
    float* o = a.col + (((size_t)n * a.C + c0) * 9 + k) * HoWo + pix;
    for (int c = c0; c < c1; ++c, p += HW, o += (size_t)9 * HoWo) {
        const float v00 = ok00 ? p[o00] : 0.f, v01 = ok01 ? p[o00 + 1] : 0.f;
        const float v10 = ok10 ? p[o00 + a.W] : 0.f, v11 = ok11 ? p[o00 + a.W + 1] : 0.f;
        // the same expression tree as the per-channel form: wy0 (wx0 v00 + wx1 v01) + wy1 (wx0 v10 + wx1 v11), then the mask
        float v = wy0 * (wx0 * v00 + b.wx1 * v01) + b.wy1 * (wx0 * v10 + b.wx1 * v11);
        if (a.mask) v *= m;
        o[0] = b.in ? v : 0.f;
    }
}

// ---- dx by GATHER (round 3): the atomic scatter costs 124 of the backward kernel's 197 us at config 5 (measured with the atomics
// compiled out: 56 us).  The geometry is shared by all channels, so it is inverted once per convolution: the thread of an (n, tap,
// output pixel) sample appends (pixel, mask * bilinear weight) to the list of each of the (<= 4) input cells its sample touches --
// lists per (cell, tap) with kDcnSlots slots, [tap][slot][cell] so that neighbouring cells' slots are neighbouring words and, with
// smooth offsets, point at neighbouring pixels of one d col plane (coalesced gathers).  A contribution that finds its list full (a
// sampling grid compressed more than 2x) is added by atomics over the channels on the spot: exact for any offsets.  Then
// dcn_gather_kernel, thread per (n, cell, 4 channels), sums weight * d col through its lists in fp64 (the slot order within a list is
// the arrival order: the fp64 sum of <= 72 fp32 products rounds to the same fp32 value whatever the order, up to ties) and adds what the
// spills left in dx.
__device__ __forceinline__ void dcn_append(const DcnArgs& a, int n, int k, int pix, int y0, int x0, float wy1, float wx1, float m) {
    const int HoWo = a.Ho * a.Wo, HW = a.H * a.W;
    const float wy[2] = {1.f - wy1, wy1}, wx[2] = {1.f - wx1, wx1};
    int* cnt = a.cnt + ((size_t)n * 9 + k) * HW;
    int2* ent = a.ent + ((size_t)n * 9 + k) * kDcnSlots * HW;
    #pragma unroll
    for (int dy = 0; dy < 2; ++dy)
        #pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int y = y0 + dy, x = x0 + dx;
            if (y < 0 || y >= a.H || x < 0 || x >= a.W) continue;
            const int cell = y * a.W + x;
            const float w = m * wy[dy] * wx[dx];
            if (w == 0.f) continue;   // (a corner without weight -- three of four on the regular grid of zero offsets -- contributes nothing: no list entry, no atomic)
            const int slot = atomicAdd(cnt + cell, 1);
            if (slot < kDcnSlots) ent[(size_t)slot * HW + cell] = make_int2(pix, __float_as_int(w));
            else {   // list full: this contribution goes the atomic way, channel by channel
                for (int c = 0; c < a.C; ++c)
                    unsafeAtomicAdd(a.dx + ((size_t)n * a.C + c) * HW + cell, w * a.dcol[(((size_t)n * a.C + c) * 9 + k) * HoWo + pix]);
            }
        }
}

// thread per (n, tap, column, group of R output rows) and channel slice (blockIdx.y): loops over the slice's channels; dx by atomic
// scatter, d offset / d mask summed in registers (one atomic per slice when the channels are split to fill the GPU on the small stages).
// The kernel is bound by the L2's float atomics, and their cost is per 64-byte REQUEST, not per lane (measured: an 8 x 8 pixel tile per
// wave with 81 lane-atomics per 64 lanes instead of 130 ran 1.7x slower -- 18 requests per channel against 10).  So the work is arranged
// to issue few, full requests:
//   * lanes are neighbouring output pixels of one row; with smooth offsets lane L+1's top-left cell IS lane L's top-right cell: lane L
//     hands its two right-column contributions to lane L+1 (one DPP shift each), which adds them to its own left-column ones;
//   * a lane walks R = 4 vertically adjacent pixels; where row r+1's top-left cell IS row r's bottom-left cell, row r's bottom
//     contributions are carried in registers into row r+1's top ones: R + 1 row requests per channel instead of 2 R.
// Any offsets stay exact: the hand-overs happen only where the integer cell coordinates coincide, and a slot that holds exactly
// zero issues nothing.
#ifndef LGD_DCN_ROWS
#define LGD_DCN_ROWS 4
#endif
constexpr int kDcnRows = LGD_DCN_ROWS;
struct DcnPix {
    float m, wy0, wy1, wx0, wx1, gy, gx, gm;
    long long o00;
    int y0, x0, pix;
    bool in, ok00, ok01, ok10, ok11, takeL, giveR, live, down;   // down: this row's bottom cells are the next row's top cells
};
template <bool DX>   // DX = false: d offset / d mask only (dx comes from the gather kernels below)
__global__ __launch_bounds__(256) void dcn_col2im_kernel(DcnArgs a) {
    constexpr int R = kDcnRows;
    const int HoWo = a.Ho * a.Wo, RG = (a.Ho + R - 1) / R;
    const long long total = (long long)a.N * 9 * RG * a.Wo;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool alive = i0 < total;
    const long long i = alive ? i0 : total - 1;      // no early exit: the lanes exchange values below
    const int wo = (int)(i % a.Wo), rg = (int)((i / a.Wo) % RG), k = (int)((i / ((long long)a.Wo * RG)) % 9);
    const int n = (int)(i / ((long long)a.Wo * RG * 9));
    const int ky = k / 3, kx = k % 3;
    const int lane = threadIdx.x & 63;
    const unsigned ln = wave_shr1((unsigned)n);
    DcnPix q[R];
    #pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ho = rg * R + r;
        DcnPix& t = q[r];
        t.live = alive && ho < a.Ho;
        t.pix = t.live ? ho * a.Wo + wo : 0;
        const float* off = a.offset + (size_t)n * a.off_bs + t.pix;
        t.m = dcn_mask(a, n, k, t.pix);
        const float py = (float)(ho * a.stride - a.pad + ky * a.dil) + off[(size_t)(2 * k) * HoWo];
        const float pxx = (float)(wo * a.stride - a.pad + kx * a.dil) + off[(size_t)(2 * k + 1) * HoWo];
        const Bilin b = bilin_setup(py, pxx, a.H, a.W);
        t.in = t.live && b.in;
        t.y0 = b.y0; t.x0 = b.x0; t.wy1 = b.wy1; t.wx1 = b.wx1; t.wy0 = 1.f - b.wy1; t.wx0 = 1.f - b.wx1;
        const bool y0ok = t.in && b.y0 >= 0 && b.y0 < a.H, y1ok = t.in && b.y0 + 1 >= 0 && b.y0 + 1 < a.H;
        const bool x0ok = b.x0 >= 0 && b.x0 < a.W, x1ok = b.x0 + 1 >= 0 && b.x0 + 1 < a.W;
        t.ok00 = y0ok && x0ok; t.ok01 = y0ok && x1ok; t.ok10 = y1ok && x0ok; t.ok11 = y1ok && x1ok;
        t.o00 = (long long)b.y0 * a.W + b.x0;          // may be negative where the sample hangs over the border
        // lane L+1 takes over lane L's right column iff it samples the same image, the same rows and the column one to the right
        // (the shifts as statements of their own: a DPP move that ends up under a lane mask reads 0 from the masked-off source lanes)
        const unsigned ly = wave_shr1((unsigned)b.y0), lx = wave_shr1((unsigned)b.x0), lin = wave_shr1(t.in ? 1u : 0u);
        t.takeL = lane > 0 && t.in && lin && ln == (unsigned)n && ly == (unsigned)b.y0 && lx + 1u == (unsigned)b.x0;
        const unsigned rtake = wave_shl1(t.takeL ? 1u : 0u);
        t.giveR = (lane < 63) & (rtake != 0u);
        t.gy = t.gx = t.gm = 0.f;
        if constexpr (!DX) {   // the contribution lists of the gather path are built here, once (channel slice 0)
            if (blockIdx.y == 0 && t.in) dcn_append(a, n, k, t.pix, b.y0, b.x0, b.wy1, b.wx1, t.m);
        }
    }
    #pragma unroll
    for (int r = 0; r < R; ++r)
        q[r].down = r + 1 < R && q[r].in && q[r + 1].in && q[r + 1].y0 == q[r].y0 + 1 && q[r + 1].x0 == q[r].x0;
    const int c0 = blockIdx.y * a.cchunk, c1 = min(a.C, c0 + a.cchunk);
    for (int c = c0; c < c1; ++c) {                                // wave-uniform trip count: every lane runs the exchanges
        const size_t plane = ((size_t)n * a.C + c) * a.H * a.W;
        const float* p = a.x + plane;
        float* dp = a.dx + plane;
        const float* dc = a.dcol + (((size_t)n * a.C + c) * 9 + k) * HoWo;
        float g[R], v00[R], v01[R], v10[R], v11[R];
        #pragma unroll
        for (int r = 0; r < R; ++r) {                              // all loads of the channel first
            const DcnPix& t = q[r];
            g[r] = t.in ? dc[t.pix] : 0.f;
            v00[r] = t.ok00 ? p[t.o00] : 0.f; v01[r] = t.ok01 ? p[t.o00 + 1] : 0.f;
            v10[r] = t.ok10 ? p[t.o00 + a.W] : 0.f; v11[r] = t.ok11 ? p[t.o00 + a.W + 1] : 0.f;
        }
        float c10 = 0.f, c11 = 0.f;                                // bottom contributions carried into the next row's top cells
        #pragma unroll
        for (int r = 0; r < R; ++r) {
            DcnPix& t = q[r];
            const float gmk = g[r] * t.m;
            float a00 = t.ok00 ? gmk * t.wy0 * t.wx0 : 0.f, a01 = t.ok01 ? gmk * t.wy0 * t.wx1 : 0.f;
            float a10 = t.ok10 ? gmk * t.wy1 * t.wx0 : 0.f, a11 = t.ok11 ? gmk * t.wy1 * t.wx1 : 0.f;
            if constexpr (DX) {   // right column -> right neighbour's left column.  Selects, not branches
                const float r01 = wave_shr1(a01), r11 = wave_shr1(a11);
                a00 += t.takeL ? r01 : 0.f;
                a10 += t.takeL ? r11 : 0.f;
                a01 = t.giveR ? 0.f : a01;
                a11 = t.giveR ? 0.f : a11;
            }
            if constexpr (DX) {
                a00 += c10; a01 += c11;                            // the row above's bottom cells (zero unless they coincide with these)
                if (t.ok00 && a00 != 0.f) unsafeAtomicAdd(dp + t.o00, a00);
                if (t.ok01 && a01 != 0.f) unsafeAtomicAdd(dp + t.o00 + 1, a01);
                if (t.down) { c10 = a10; c11 = a11; }
                else {
                    c10 = c11 = 0.f;
                    if (t.ok10 && a10 != 0.f) unsafeAtomicAdd(dp + t.o00 + a.W, a10);
                    if (t.ok11 && a11 != 0.f) unsafeAtomicAdd(dp + t.o00 + a.W + 1, a11);
                }
            }
            t.gy += gmk * (t.wx0 * (v10[r] - v00[r]) + t.wx1 * (v11[r] - v01[r]));
            t.gx += gmk * (t.wy0 * (v01[r] - v00[r]) + t.wy1 * (v11[r] - v10[r]));
            t.gm += g[r] * (t.wy0 * (t.wx0 * v00[r] + t.wx1 * v01[r]) + t.wy1 * (t.wx0 * v10[r] + t.wx1 * v11[r]));
        }
    }
    #pragma unroll
    for (int r = 0; r < R; ++r) {
        const DcnPix& t = q[r];
        if (!t.live) continue;
        float* oy = a.doffset + (size_t)n * a.doff_bs + (size_t)(2 * k) * HoWo + t.pix;
        float* om = a.dmask ? a.dmask + (size_t)n * a.dmask_bs + (size_t)k * HoWo + t.pix : nullptr;
        const float gm = a.sig ? t.gm * (t.m * (1.f - t.m)) : t.gm;   // packed form: gradient of the mask LOGIT
        if (gridDim.y == 1) {
            oy[0] = t.gy; oy[HoWo] = t.gx;
            if (om) om[0] = gm;
        } else if (t.in) {  // outputs zeroed by the host entry
            unsafeAtomicAdd(oy, t.gy); unsafeAtomicAdd(oy + HoWo, t.gx);
            if (om) unsafeAtomicAdd(om, gm);
        }
    }
}

constexpr int kDcnGatherCh = 4;   // channels per thread (measured 4 / 8 / 16: 62 / 72 / 170 us per launch at config 5; fp32 sums: 4 us less)
typedef double dcn_acc_t;
__global__ __launch_bounds__(256) void dcn_gather_kernel(DcnArgs a) {
    const int HoWo = a.Ho * a.Wo, HW = a.H * a.W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.N * HW) return;
    const int cell = (int)(i % HW), n = (int)(i / HW);
    const int c0 = blockIdx.y * kDcnGatherCh, nc = min(kDcnGatherCh, a.C - c0);
    dcn_acc_t acc[kDcnGatherCh];
    #pragma unroll
    for (int j = 0; j < kDcnGatherCh; ++j) acc[j] = 0;
    const float* dc = a.dcol + ((size_t)n * a.C + c0) * 9 * HoWo;
    for (int k = 0; k < 9; ++k) {
        const int cn = min(a.cnt[((size_t)n * 9 + k) * HW + cell], kDcnSlots);
        const int2* ent = a.ent + ((size_t)n * 9 + k) * kDcnSlots * HW + cell;
        // all slots of the tap, then all their d col values, before anything is summed: a loop with the list's own trip count is a chain
        // of two dependent loads per entry (111 us per launch at config 5; predicated and unrolled: see DESIGN section 9)
        int2 en[kDcnSlots];
        #pragma unroll
        for (int e = 0; e < kDcnSlots; ++e) en[e] = e < cn ? ent[(size_t)e * HW] : make_int2(0, 0);
        #pragma unroll
        for (int e = 0; e < kDcnSlots; ++e) {
            const float w = __int_as_float(en[e].y);          // empty slot: weight 0, pixel 0
            const float* src = dc + (size_t)k * HoWo + en[e].x;
            float v[kDcnGatherCh];
            #pragma unroll
            for (int j = 0; j < kDcnGatherCh; ++j) v[j] = (e < cn && j < nc) ? src[(size_t)j * 9 * HoWo] : 0.f;
            #pragma unroll
            for (int j = 0; j < kDcnGatherCh; ++j) acc[j] += (dcn_acc_t)(w * v[j]);
        }
    }
    float* o = a.dx + ((size_t)n * a.C + c0) * HW + cell;
    #pragma unroll
    for (int j = 0; j < kDcnGatherCh; ++j)
        if (j < nc) o[(size_t)j * HW] += (float)acc[j];   // + what full lists spilled (zero-initialised by the entry point)
}

static int dcn_fill(DcnArgs& a, const float* x, const float* offset, const float* mask, int N, int C, int H, int W, int stride,
                    int pad, int dil) {
    if (!x || !offset || N < 1 || C < 1 || H < 1 || W < 1 || stride < 1 || dil < 1 || pad < 0) return LGD_EINVAL;
    a.x = x; a.offset = offset; a.mask = mask; a.N = N; a.C = C; a.H = H; a.W = W; a.stride = stride; a.pad = pad; a.dil = dil;
    a.Ho = (H + 2 * pad - dil * 2 - 1) / stride + 1;
    a.Wo = (W + 2 * pad - dil * 2 - 1) / stride + 1;
    a.col = nullptr; a.dcol = nullptr; a.dx = nullptr; a.doffset = nullptr; a.dmask = nullptr; a.cchunk = C; a.cnt = nullptr; a.ent = nullptr;
    if (a.Ho < 1 || a.Wo < 1) return LGD_EINVAL;
    const long long HoWo = (long long)a.Ho * a.Wo;
    a.off_bs = a.doff_bs = 18 * HoWo; a.mask_bs = a.dmask_bs = 9 * HoWo; a.sig = 0;
    return LGD_OK;
}

}  // namespace lgd

extern "C" {

static int dcn_im2col_launch(lgd::DcnArgs& a, void* stream) {
    const int N = a.N, C = a.C;
    const long long total = (long long)N * 9 * a.Ho * a.Wo;
    // split the channel loop until ~0.5 M threads are in flight
    int slices = (int)((500000 + total - 1) / total);
    slices = slices < 1 ? 1 : (slices > C / 8 ? (C / 8 > 0 ? C / 8 : 1) : slices);
    a.cchunk = (C + slices - 1) / slices;
    slices = (C + a.cchunk - 1) / a.cchunk;
    LGD_LAUNCH("dcn_im2col_kernel", lgd::dcn_im2col_kernel, dim3((unsigned)((total + 255) / 256), slices), dim3(256), 0, (hipStream_t)stream, a);
    return lgd::check_launch();
}

int lgd_dcn_im2col(const float* x, const float* offset, const float* mask, int N, int C, int H, int W, int stride, int pad,
                   int dilation, float* col, void* stream) {
    lgd::DcnArgs a;
    if (!col || lgd::dcn_fill(a, x, offset, mask, N, C, H, W, stride, pad, dilation) != LGD_OK) return LGD_EINVAL;
    a.col = col;
    return dcn_im2col_launch(a, stream);
}

// packed form: om (N, 27, Ho, Wo) is the offset convolution's own output -- channels 0..17 the offsets, 18..26 the mask LOGITS
static void dcn_pack(lgd::DcnArgs& a, const float* om) {
    const long long HoWo = (long long)a.Ho * a.Wo;
    a.mask = om + 18 * HoWo;
    a.off_bs = a.mask_bs = a.doff_bs = a.dmask_bs = 27 * HoWo;
    a.sig = 1;
}

int lgd_dcn_im2col_packed(const float* x, const float* om, int N, int C, int H, int W, int stride, int pad, int dilation, float* col,
                          void* stream) {
    lgd::DcnArgs a;
    if (!col || lgd::dcn_fill(a, x, om, om, N, C, H, W, stride, pad, dilation) != LGD_OK) return LGD_EINVAL;
    dcn_pack(a, om);
    a.col = col;
    return dcn_im2col_launch(a, stream);
}

size_t lgd_dcn_ws_bytes(int N, int H, int W) {
    if (N < 1 || H < 1 || W < 1) return 0;
    return (size_t)N * 9 * H * W * (sizeof(int) + lgd::kDcnSlots * sizeof(int2)) + 16;
}

static int dcn_col2im_launch(lgd::DcnArgs& a, void* ws, void* stream) {
    const int N = a.N, C = a.C, H = a.H, W = a.W;
    float* dx = a.dx; float* doffset = a.doffset; float* dmask = a.dmask;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)N * 9 * ((a.Ho + lgd::kDcnRows - 1) / lgd::kDcnRows) * a.Wo;   // a thread walks kDcnRows output rows
    // split the channel loop until ~0.5 M threads are in flight (res5 has only 19 K (n, tap, pixel) triples)
    int slices = (int)((500000 + total - 1) / total);
    slices = slices < 1 ? 1 : (slices > C / 8 ? (C / 8 > 0 ? C / 8 : 1) : slices);
    a.cchunk = (C + slices - 1) / slices;
    slices = (C + a.cchunk - 1) / a.cchunk;
    // what has to start at zero: dx (spills of full lists / the atomic path), d offset and d mask when the channel loop is split, the list
    // counters.  A caller that carves them in this order out of ONE allocation (gaps under 16 bytes: alignment padding) gets one fill
    // instead of four -- at 2 images per GPU a fill is a 5 us launch, 120 of them per step of the R-101-DCNv2 student
    const size_t HW = (size_t)H * W;
    struct Zero { char* p; size_t n; } z[4];
    int nz = 0;
    z[nz++] = {reinterpret_cast<char*>(dx), (size_t)N * C * HW * sizeof(float)};
    if (slices > 1 && a.sig) z[nz++] = {reinterpret_cast<char*>(doffset), (size_t)N * 27 * a.Ho * a.Wo * sizeof(float)};   // packed: one tensor
    else if (slices > 1) {
        z[nz++] = {reinterpret_cast<char*>(doffset), (size_t)N * 18 * a.Ho * a.Wo * sizeof(float)};
        if (a.dmask) z[nz++] = {reinterpret_cast<char*>(dmask), (size_t)N * 9 * a.Ho * a.Wo * sizeof(float)};
    }
    if (ws) z[nz++] = {reinterpret_cast<char*>(ws), (size_t)N * 9 * HW * sizeof(int)};
    for (int i = 0; i < nz;) {
        Zero m = z[i++];
        while (i < nz && z[i].p >= m.p + m.n && (size_t)(z[i].p - (m.p + m.n)) < 16) { m.n = (size_t)(z[i].p - m.p) + z[i].n; ++i; }
        if (hipMemsetAsync(m.p, 0, m.n, st) != hipSuccess) return LGD_ELAUNCH;
    }
    const dim3 grid((unsigned)((total + 255) / 256), slices);
    if (!ws) {   // no list workspace: dx by atomic scatter inside the same kernel (round 2's path, kept for A/B runs)
        LGD_LAUNCH("dcn_col2im_kernel", lgd::dcn_col2im_kernel<true>, grid, dim3(256), 0, st, a);
        return lgd::check_launch();
    }
    a.cnt = reinterpret_cast<int*>(ws);
    a.ent = reinterpret_cast<int2*>(reinterpret_cast<char*>(ws) + (((size_t)N * 9 * HW * sizeof(int) + 15) & ~(size_t)15));
    LGD_LAUNCH("dcn_col2im_kernel", lgd::dcn_col2im_kernel<false>, grid, dim3(256), 0, st, a);   // d offset, d mask + the lists
    LGD_LAUNCH("dcn_gather_kernel", lgd::dcn_gather_kernel, dim3((unsigned)(((long long)N * HW + 255) / 256), (C + lgd::kDcnGatherCh - 1) / lgd::kDcnGatherCh),
               dim3(256), 0, st, a);
    return lgd::check_launch();
}

int lgd_dcn_col2im(const float* x, const float* offset, const float* mask, const float* dcol, int N, int C, int H, int W,
                   int stride, int pad, int dilation, float* dx, float* doffset, float* dmask, void* ws, void* stream) {
    lgd::DcnArgs a;
    if (!dcol || !dx || !doffset || (mask && !dmask) || lgd::dcn_fill(a, x, offset, mask, N, C, H, W, stride, pad, dilation) != LGD_OK)
        return LGD_EINVAL;
    a.dcol = dcol; a.dx = dx; a.doffset = doffset; a.dmask = mask ? dmask : nullptr;
    return dcn_col2im_launch(a, ws, stream);
}

int lgd_dcn_col2im_packed(const float* x, const float* om, const float* dcol, int N, int C, int H, int W, int stride, int pad,
                          int dilation, float* dx, float* dom, void* ws, void* stream) {
    lgd::DcnArgs a;
    if (!dcol || !dx || !dom || lgd::dcn_fill(a, x, om, om, N, C, H, W, stride, pad, dilation) != LGD_OK) return LGD_EINVAL;
    dcn_pack(a, om);
    a.dcol = dcol; a.dx = dx; a.doffset = dom; a.dmask = dom + 18 * (long long)a.Ho * a.Wo;
    return dcn_col2im_launch(a, ws, stream);
}

}  // extern "C"
