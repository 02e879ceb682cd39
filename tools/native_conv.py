"""The 3x3 convolutions issued as ONE native call per direction (csrc/conv.hip: lgd_conv3x3_fwd / lgd_conv3x3_bwd -- filter
transforms, data transforms and the rocBLAS channel GEMMs all launched from inside the library): the entry points a host WITHOUT a
tensor library binds, here wired into the same autograd nodes the product uses so that the two issue paths can be
compared in one process.

    python tools/native_conv.py [yaml] [batch]      in-call A/B of the training step: product (Python composes the pipeline around
                                                     torch.bmm) vs these nodes; prints ms/step and the host issue time of both

Measured on MI355X (profiles/r03_native_conv_ab.txt): the step does not get faster -- it is GPU-bound (kernel time = step time to 1 %)
-- and the host issue time goes UP (config 4: 21.9 -> 32.5 ms/step): filling a 1 KB ctypes structure costs more Python than the five
ctypes / torch calls it replaces.  The product therefore keeps composing the pipeline itself; tests/test_kernels_gpu.py holds the
native entry points to the same results."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from lgd_amd import hip, ops  # noqa: E402
from lgd_amd.ops import _WINO_MASK_DTYPE, _count_bytes  # noqa: E402

_GEMM_FLOPS = ops._GEMM_FLOPS


def _timer_on():
    return ops._TIMER_ON


_GEMM_SOLUTIONS = None


def _gemm_solution(kind, Ct, Ci, T, nf):
    """rocBLAS solution index for one of the three channel products of a Winograd convolution (csrc/conv.hip::wino_gemm), from the
    same table torch's TunableOp reads (lgd_amd/tuning/tunableop_gfx950.csv: measured on an MI355X, lookup only).  0 = rocBLAS's own
    choice: shape not in the table, the table's best entry is a hipBLASLt solution, or the table was tuned for another rocBLAS build."""
    global _GEMM_SOLUTIONS
    if _GEMM_SOLUTIONS is None:
        _GEMM_SOLUTIONS = {}
        path = os.path.join(ROOT, "lgd_amd", "tuning", "tunableop_gfx950.csv")
        if os.path.exists(path) and os.environ.get("LGD_TUNED_GEMM", "1") != "0":
            buf = ctypes.create_string_buffer(256)
            have = buf.value.decode() if hip.load().lgd_blas_version(buf, 256) == 0 else ""
            rows, want = [], None
            for line in open(path):
                f = line.strip().split(",")
                if f[0] == "Validator" and f[1] == "ROCBLAS_VERSION":
                    want = f[2]
                elif len(f) >= 3 and f[2].startswith("Gemm_Rocblas_"):
                    rows.append((f[1], int(f[2][len("Gemm_Rocblas_"):])))
            if want is not None and want == have:
                _GEMM_SOLUTIONS = dict(rows)
    ld = nf * T
    key = ("nn_%d_%d_%d_B_%d_ld_%d_%d_%d" % (T, Ct, Ci, nf, ld, Ci, ld) if kind == 0 else
           "nn_%d_%d_%d_B_%d_ld_%d_%d_%d" % (T, Ci, Ct, nf, ld, Ct, ld) if kind == 1 else
           "tn_%d_%d_%d_B_%d_ld_%d_%d_%d" % (Ci, Ct, T, nf, ld, ld, Ci))
    return _GEMM_SOLUTIONS.get(key, 0)


def _ptr_or_none(t):
    return t.data_ptr() if t is not None else None


class NativeConv3x3K(torch.autograd.Function):
    """K filters nn.Conv2d(Ci, Co_k, 3, stride 1, padding 1) [+ ReLU] applied to the SAME L maps (the pyramid levels; K = 1: one
    conv, K = 2: e.g. the first convs of the cls / bbox towers, which read the same features) in the minimal-filtering form
    F(tile x tile, 3x3), tile = 6 or 4, as ONE native call per direction (csrc/conv.hip: lgd_conv3x3_fwd / lgd_conv3x3_bwd): HIP data
    transforms around per-frequency channel GEMMs (rocBLAS fp32 MFMA, issued by the library with the solution index of the tuning
    table) over the concatenated tiles of all levels.  The input is transformed ONCE for all K filters (their U are stacked along
    C_out: one GEMM), and the backward sums their input gradients inside the dV GEMM (K = sum Co_k) -- one adjoint input transform, no
    gradient-accumulation pass.  Forward, input gradient and weight gradient all run at 64/324 (tile 6) or 1/4 (tile 4) of the direct
    multiplies.  The backward is the autograd of the pipeline itself: dy is expanded ONCE (dM = A dy A^T), dV[f] = U[f]^T dM[f] comes
    back through the adjoint of the input transform, the weight gradient is dU[f] = dM[f] V[f]^T.
    apply(K, relu, tile, scales, pre, w_1, b_1, ..., w_K, b_K, x_1, ..., x_L) -> K * L maps, filter-major; scales: None or one per-output-
    channel factor (a buffer, no gradient) per filter, applied to the filter inside its transform; pre: None, or a per-INPUT-
    channel bias (a buffer): the maps are then pre-activations and the convolution runs on relu(x + pre[c]) -- the bias + ReLU epilogue
    of the producing 1x1 convolution folded into the input transform, its backward mask into the adjoint transform, so the
    gradient returned for x is the gradient of the RAW map."""

    @staticmethod
    def forward(ctx, K, relu, tile, scales, pre, *args):
        ws, bs, xs = list(args[0:2 * K:2]), list(args[1:2 * K:2]), list(args[2 * K:])
        if tile not in _WINO_MASK_DTYPE:
            raise hip.LgdHipError("Winograd output tile must be 4 or 6")
        if K > hip.MAX_FILTERS or len(xs) > hip.MAX_LEVELS:
            raise hip.LgdHipError("at most %d filters on %d maps per call" % (hip.MAX_FILTERS, hip.MAX_LEVELS))
        scales = list(scales) if scales is not None else [None] * K
        hip.require_gpu(*ws, *xs)
        lib = hip.load()
        ws = [hip.dense_f32(w) for w in ws]
        xs = [hip.dense_f32(x) for x in xs]
        bs = [hip.dense_f32(b) if b is not None else None for b in bs]
        L, N, Ci = len(xs), xs[0].shape[0], xs[0].shape[1]
        Cos = [w.shape[0] for w in ws]
        Ct = sum(Cos)
        dev = ws[0].device
        nf = (tile + 2) ** 2
        mdt, mb = _WINO_MASK_DTYPE[tile], _WINO_MASK_DTYPE[tile].itemsize
        a = hip.Conv3x3FwdArgs()
        a.L, a.N, a.Ci, a.K, a.tile, a.relu = L, N, Ci, K, tile, int(relu)
        for i, x in enumerate(xs):
            a.level_hw[2 * i], a.level_hw[2 * i + 1] = x.shape[2], x.shape[3]
            a.x[i] = x.data_ptr()
        T = lib.lgd_wino_tiles(a.level_hw, L, N, tile)
        need_x = any(ctx.needs_input_grad[5 + 2 * K:])
        need_w = any(ctx.needs_input_grad[5:5 + 2 * K:2])
        U = torch.empty((nf, Ct, Ci), dtype=torch.float32, device=dev)
        Ut = torch.empty((nf, Ci, Ct), dtype=torch.float32, device=dev)
        V = torch.empty((Ci, nf, T), dtype=torch.float32, device=dev)
        M = torch.empty((Ct, nf, T), dtype=torch.float32, device=dev)
        pre = hip.dense_f32(pre) if pre is not None else None
        pre_bits = torch.empty((Ci, T), dtype=mdt, device=dev) if pre is not None and need_x else None
        # ReLU mask for the backward: one bit per pixel, a table entry per tile, written by the output transform, so the backward
        # reads 1 bit instead of 4 bytes per pixel and the forward output is not kept alive
        bits = torch.empty((Ct, T), dtype=mdt, device=dev) if relu else None
        ys = []
        for k in range(K):
            a.Co[k] = Cos[k]
            a.w[k] = ws[k].data_ptr()
            a.scale[k] = _ptr_or_none(scales[k])
            a.bias[k] = _ptr_or_none(bs[k])
            for i, x in enumerate(xs):
                y = torch.empty((N, Cos[k], x.shape[2], x.shape[3]), dtype=torch.float32, device=dev)
                a.y[k * L + i] = y.data_ptr()
                ys.append(y)
        a.pre_bias, a.pre_bits, a.relu_bits = _ptr_or_none(pre), _ptr_or_none(pre_bits), _ptr_or_none(bits)
        a.U, a.Ut, a.V, a.M = U.data_ptr(), Ut.data_ptr(), V.data_ptr(), M.data_ptr()
        a.sol_fwd = _gemm_solution(0, Ct, Ci, T, nf)
        hip.check(lib.lgd_conv3x3_fwd(ctypes.byref(a), hip.stream_ptr()), "lgd_conv3x3_fwd")
        px = 4 * N * sum(x.shape[2] * x.shape[3] for x in xs)  # bytes of one channel of the maps
        fb = 4 * nf * T                                        # bytes of one channel of a frequency buffer
        if _timer_on():
            _count_bytes("wino_in_kernel", (px + fb) * Ci)
            for k in range(K):
                _count_bytes("wino_out_kernel", (px + fb + (mb * T if bits is not None else 0)) * Cos[k])
            _GEMM_FLOPS["wino_gemm_fwd"] = _GEMM_FLOPS.get("wino_gemm_fwd", 0) + 2 * nf * Ct * Ci * T
        ctx.save_for_backward(Ut, V if need_w else None, bits, pre_bits)   # the backward needs U^T (dV = U^T dM)
        ctx.scales = scales
        ctx.meta = (K, L, N, Ci, Cos, [int(v) for v in a.level_hw[:2 * L]], T, [b is not None for b in bs], [tuple(x.shape[2:]) for x in xs],
                    tile, px, fb)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        Ut, V, bits, pre_bits = ctx.saved_tensors
        K, L, N, Ci, Cos, hw, T, has_bias, shapes, tile, px, fb = ctx.meta
        Ct = sum(Cos)
        lib = hip.load()
        dev = Ut.device
        nf = (tile + 2) ** 2
        mb = _WINO_MASK_DTYPE[tile].itemsize
        need_ws = list(ctx.needs_input_grad[5:5 + 2 * K:2])
        need_bs = [hb and nb for hb, nb in zip(has_bias, ctx.needs_input_grad[6:6 + 2 * K:2])]
        need_w, need_x = any(need_ws), any(ctx.needs_input_grad[5 + 2 * K:])
        dws, dbs, dxs = [None] * K, [None] * K, [None] * L
        if not (need_x or need_w or any(need_bs)):
            return (None, None, None, None, None, *[None] * (2 * K), *dxs)
        a = hip.Conv3x3BwdArgs()
        a.L, a.N, a.Ci, a.K, a.tile, a.dM_ready = L, N, Ci, K, tile, 0
        for i, v in enumerate(hw):
            a.level_hw[i] = v
        keep = []
        for i, g in enumerate(dys):   # an output nothing downstream used arrives as None
            g = hip.dense_f32(g) if g is not None else torch.zeros((N, Cos[i // L]) + shapes[i % L], dtype=torch.float32, device=dev)
            keep.append(g)
            a.dy[i] = g.data_ptr()
        dM = torch.empty((Ct, nf, T), dtype=torch.float32, device=dev)
        a.relu_bits, a.dM, a.Ut = _ptr_or_none(bits), dM.data_ptr(), Ut.data_ptr()
        if need_x:
            dV = torch.empty((Ci, nf, T), dtype=torch.float32, device=dev)
            a.dV = dV.data_ptr()
            dxs = [torch.empty((N, Ci) + s_, dtype=torch.float32, device=dev) for s_ in shapes]
            for i, t in enumerate(dxs):
                a.dx[i] = t.data_ptr()
            a.pre_bits = _ptr_or_none(pre_bits)
            a.sol_dx = _gemm_solution(1, Ct, Ci, T, nf)
        if need_w:
            dU = torch.empty((nf, Ct, Ci), dtype=torch.float32, device=dev)
            a.V, a.dU = V.data_ptr(), dU.data_ptr()
            a.sol_dw = _gemm_solution(2, Ct, Ci, T, nf)
        for k in range(K):
            a.Co[k] = Cos[k]
            a.scale[k] = _ptr_or_none(ctx.scales[k])
            if need_ws[k]:
                dws[k] = torch.empty((Cos[k], Ci, 3, 3), dtype=torch.float32, device=dev)
                a.dw[k] = dws[k].data_ptr()
            if need_bs[k]:
                dbs[k] = torch.empty((Cos[k],), dtype=torch.float32, device=dev)
                a.db[k] = dbs[k].data_ptr()
        hip.check(lib.lgd_conv3x3_bwd(ctypes.byref(a), hip.stream_ptr()), "lgd_conv3x3_bwd")
        if _timer_on():
            for k in range(K):
                _count_bytes("wino_out_t_kernel", (px + fb + (mb * T if bits is not None else 0)) * Cos[k])
            if need_x:
                _count_bytes("wino_in_t_kernel", (px + fb) * Ci)
                _GEMM_FLOPS["wino_gemm_dx"] = _GEMM_FLOPS.get("wino_gemm_dx", 0) + 2 * nf * Ct * Ci * T
            if need_w:
                _GEMM_FLOPS["wino_gemm_dw"] = _GEMM_FLOPS.get("wino_gemm_dw", 0) + 2 * nf * Ct * Ci * T
        return (None, None, None, None, None, *[g for pair in zip(dws, dbs) for g in pair], *dxs)


class NativeConv3x3Chain(torch.autograd.Function):
    """K convolutions 3x3 / stride 1 / padding 1 in SEQUENCE over the same L maps, conv k [+ ReLU if relus[k]] feeding conv k+1 and
    nothing else (the head towers after their first conv incl. the score conv, the adapter: distillator.py:107-109 ->
    retinanet.py:36-43, sequential_convs.py:10-12).  The forward is the per-conv Winograd pipeline of _Conv3x3K (one native call per
    conv).  The backward keeps the gradient in the FREQUENCY domain across a link: dV_k = U_k^T dM_k goes through ONE kernel
    (lgd_wino_in_t_out_t: adjoint input transform, ReLU mask of conv k-1, A . A^T) into dM_{k-1}; the intermediate gradient maps are
    neither written nor re-read.  apply(K, relus, tile, w_1, b_1, ..., w_K, b_K, x_1, ..., x_L) -> the L maps of the last conv."""

    @staticmethod
    def forward(ctx, K, relus, tile, *args):
        ws, bs, xs = list(args[0:2 * K:2]), list(args[1:2 * K:2]), list(args[2 * K:])
        hip.require_gpu(*ws, *xs)
        lib = hip.load()
        nf = (tile + 2) ** 2
        mdt, mb = _WINO_MASK_DTYPE[tile], _WINO_MASK_DTYPE[tile].itemsize
        ws = [hip.dense_f32(w) for w in ws]
        xs = [hip.dense_f32(x) for x in xs]
        bs = [hip.dense_f32(b) if b is not None else None for b in bs]
        L, N = len(xs), xs[0].shape[0]
        dev = ws[0].device
        shapes = [tuple(x.shape[2:]) for x in xs]
        hw = [d for s_ in shapes for d in s_]
        a = hip.Conv3x3FwdArgs()
        a.L, a.N, a.K, a.tile = L, N, 1, tile
        for i, v in enumerate(hw):
            a.level_hw[i] = v
        T = lib.lgd_wino_tiles(a.level_hw, L, N, tile)
        px = 4 * N * sum(h * w_ for h, w_ in shapes)   # bytes of one channel of the maps
        fb = 4 * nf * T                                # bytes of one channel of a frequency buffer
        need_ws = list(ctx.needs_input_grad[3:3 + 2 * K:2])
        saved, cur = [], xs
        for k in range(K):
            Co, Ci = ws[k].shape[0], ws[k].shape[1]
            U = torch.empty((nf, Co, Ci), dtype=torch.float32, device=dev)
            Ut = torch.empty((nf, Ci, Co), dtype=torch.float32, device=dev)
            V = torch.empty((Ci, nf, T), dtype=torch.float32, device=dev)
            M = torch.empty((Co, nf, T), dtype=torch.float32, device=dev)
            bits = torch.empty((Co, T), dtype=mdt, device=dev) if relus[k] else None
            nxt = [torch.empty((N, Co) + s_, dtype=torch.float32, device=dev) for s_ in shapes]
            a.Ci, a.relu = Ci, int(relus[k])
            a.Co[0], a.w[0], a.bias[0] = Co, ws[k].data_ptr(), _ptr_or_none(bs[k])
            for i in range(L):
                a.x[i], a.y[i] = cur[i].data_ptr(), nxt[i].data_ptr()
            a.relu_bits = _ptr_or_none(bits)
            a.U, a.Ut, a.V, a.M = U.data_ptr(), Ut.data_ptr(), V.data_ptr(), M.data_ptr()
            a.sol_fwd = _gemm_solution(0, Co, Ci, T, nf)
            hip.check(lib.lgd_conv3x3_fwd(ctypes.byref(a), hip.stream_ptr()), "lgd_conv3x3_fwd")
            if _timer_on():
                _count_bytes("wino_in_kernel", (px + fb) * Ci)
                _count_bytes("wino_out_kernel", (px + fb + (mb * T if bits is not None else 0)) * Co)
                _GEMM_FLOPS["wino_gemm_fwd"] = _GEMM_FLOPS.get("wino_gemm_fwd", 0) + 2 * nf * Co * Ci * T
            cur = nxt
            saved += [Ut, V if need_ws[k] else None, bits]
        ctx.save_for_backward(*saved)
        ctx.meta = (K, L, N, hw, T, shapes, [b is not None for b in bs], px, fb, tile)
        return tuple(cur)

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors
        K, L, N, hw, T, shapes, has_bias, px, fb, tile = ctx.meta
        lib = hip.load()
        nf = (tile + 2) ** 2
        mb = _WINO_MASK_DTYPE[tile].itemsize
        dev = saved[0].device
        need_ws = list(ctx.needs_input_grad[3:3 + 2 * K:2])
        need_bs = [hb and nb for hb, nb in zip(has_bias, ctx.needs_input_grad[4:4 + 2 * K:2])]
        need_x = any(ctx.needs_input_grad[3 + 2 * K:])
        dws, dbs, dxs = [None] * K, [None] * K, [None] * L
        Co = saved[3 * (K - 1)].shape[2]
        dys = [hip.dense_f32(g) if g is not None else torch.zeros((N, Co) + shapes[i], dtype=torch.float32, device=dev) for i, g in enumerate(dys)]
        a = hip.Conv3x3BwdArgs()
        a.L, a.N, a.K, a.tile = L, N, 1, tile
        for i, v in enumerate(hw):
            a.level_hw[i] = v
        dM = torch.empty((Co, nf, T), dtype=torch.float32, device=dev)
        for k in range(K - 1, -1, -1):
            Ut, V, bits = saved[3 * k], saved[3 * k + 1], saved[3 * k + 2]
            Ci, Co = Ut.shape[1], Ut.shape[2]
            last = k == K - 1
            a.Ci, a.dM_ready = Ci, 0 if last else 1
            a.Co[0] = Co
            if last:
                for i in range(L):
                    a.dy[i] = dys[i].data_ptr()
            a.relu_bits = _ptr_or_none(bits) if last else None
            a.dM, a.Ut = dM.data_ptr(), Ut.data_ptr()
            dU = None
            a.V = a.dU = None
            a.dw[0] = a.db[0] = None
            if need_ws[k]:
                dU = torch.empty((nf, Co, Ci), dtype=torch.float32, device=dev)
                dws[k] = torch.empty((Co, Ci, 3, 3), dtype=torch.float32, device=dev)
                a.V, a.dU, a.dw[0] = V.data_ptr(), dU.data_ptr(), dws[k].data_ptr()
                a.sol_dw = _gemm_solution(2, Co, Ci, T, nf)
            if need_bs[k]:
                dbs[k] = torch.empty((Co,), dtype=torch.float32, device=dev)
                a.db[0] = dbs[k].data_ptr()
            dV = dM_prev = None
            a.dV = a.dM_prev = a.prev_bits = a.pre_bits = None
            for i in range(L):
                a.dx[i] = None
            if k > 0 or need_x:
                dV = torch.empty((Ci, nf, T), dtype=torch.float32, device=dev)
                a.dV = dV.data_ptr()
                a.sol_dx = _gemm_solution(1, Co, Ci, T, nf)
                if k > 0:   # the link to conv k-1: dM_{k-1} = A (in_t(dV) . relu mask) A^T without the map in between
                    dM_prev = torch.empty((Ci, nf, T), dtype=torch.float32, device=dev)
                    a.dM_prev, a.prev_bits = dM_prev.data_ptr(), _ptr_or_none(saved[3 * (k - 1) + 2])
                else:
                    dxs = [torch.empty((N, Ci) + s_, dtype=torch.float32, device=dev) for s_ in shapes]
                    for i in range(L):
                        a.dx[i] = dxs[i].data_ptr()
            hip.check(lib.lgd_conv3x3_bwd(ctypes.byref(a), hip.stream_ptr()), "lgd_conv3x3_bwd")
            if _timer_on():
                if last:
                    _count_bytes("wino_out_t_kernel", (px + fb + (mb * T if bits is not None else 0)) * Co)
                if need_ws[k]:
                    _GEMM_FLOPS["wino_gemm_dw"] = _GEMM_FLOPS.get("wino_gemm_dw", 0) + 2 * nf * Co * Ci * T
                if dV is not None:
                    _GEMM_FLOPS["wino_gemm_dx"] = _GEMM_FLOPS.get("wino_gemm_dx", 0) + 2 * nf * Co * Ci * T
                    if k > 0:
                        _count_bytes("wino_in_t_out_t_kernel", (2 * fb + (mb * T if saved[3 * (k - 1) + 2] is not None else 0)) * Ci)
                    else:
                        _count_bytes("wino_in_t_kernel", (px + fb) * Ci)
            dM = dM_prev
        return (None, None, None, *[g for pair in zip(dws, dbs) for g in pair], *dxs)




class installed:
    """context manager: the product's convolution nodes replaced by the native-call ones"""

    def __enter__(self):
        self.prev = (ops._Conv3x3K, ops._Conv3x3Chain)
        ops._Conv3x3K, ops._Conv3x3Chain = NativeConv3x3K, NativeConv3x3Chain
        return self

    def __exit__(self, *exc):
        ops._Conv3x3K, ops._Conv3x3Chain = self.prev


if __name__ == "__main__":
    import time
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    yaml = sys.argv[1] if len(sys.argv) > 1 else "lgd_retinanet_r101"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cfg = config.setup_cfg(os.path.join(ROOT, "configs", yaml + ".yaml"), ["MODEL.DEVICE", "cuda"])
    torch.manual_seed(0)
    tr = Trainer(cfg, build_model(cfg))
    data = synthetic_batch(B, 800, 1333, 10, seed=1, device="cuda")

    def measure(tag):
        for i in range(6):
            tr.step(data, 40000 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            tr.step(data, 40010 + i)
        torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / 20
        issue = []
        for i in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.step(data, 40040 + i)
            issue.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        print("%-44s %6.2f ms/step   host issue %5.1f ms/step" % (tag, 1e3 * step, 1e3 * sum(issue) / len(issue)), flush=True)
    print("%s, %d img/GPU, 800x1333" % (yaml, B))
    for rep in range(2):
        measure("product (pipeline composed by the host)")
        with installed():
            measure("native (one lgd_conv3x3_* call per direction)")
