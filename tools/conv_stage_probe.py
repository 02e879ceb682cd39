"""Round 6 diagnosis (follows tools/fpn_race_probe.py): the p4 output convolution of the FPN (256 -> 256 over 8 maps of 50 x 84: F(6x6) transforms +
csrc/gemm3.hip) stage by stage -- filter image, V, M, y -- on the main stream while the p3 convolution (8 x 100 x 168: the f16x2 pipeline) runs on a
side stream; every stage's buffer bit-compared with the quiet run's.  Which stage is it that changes under concurrency?
    python tools/conv_stage_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lgd_amd import hip, ops  # noqa: E402

if "--lib" in sys.argv:   # a lab build (tools/coherence_lab.sh)
    i = sys.argv.index("--lib")
    hip._LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
H2 = "--h2" in sys.argv
ROUNDS = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 40
TORCH_ONLY = "--torch" in sys.argv   # the victim is a pair of torch's OWN kernels (elementwise producer -> reduction consumer): is the platform's
                                     # cross-kernel visibility under a second stream at fault for everybody, or only for this library's kernels?   # the victim on the f16x2 pipeline as well (wino6_in<H2> -> h2_fwd -> wino6_out)
dev = torch.device("cuda:0")
lib = hip.load()
g = torch.Generator(device=dev).manual_seed(3)
x4 = torch.randn((8, 256, 50, 84), device=dev, generator=g)
x3 = torch.randn((8, 256, 100, 168), device=dev, generator=g)
w4 = torch.randn((256, 256, 3, 3), device=dev, generator=g) * 0.02
w3 = torch.randn((256, 256, 3, 3), device=dev, generator=g) * 0.02
b4 = torch.randn(256, device=dev, generator=g)
hw = hip.int_array([50, 84])
T = lib.lgd_wino_tiles(hw, 1, 8, 6)
print("library:", hip.lib_path(), "| victim on", "h2" if H2 else "gemm3", "| T =", T, "| aggressor:", sys.argv[sys.argv.index("--aggressor") + 1] if "--aggressor" in sys.argv else "conv")


def staged():
    if H2:
        filt = ops._h2_filters(lib, [w4], [None], 256, dev, False)
        amax = ops._amax_bits(lib, [x4], hw)
        V, vinv = ops._h2_buf(256, T, dev), torch.empty(1, dtype=torch.float32, device=dev)
        hip.check(lib.lgd_wino_in_h2(hip.ptr_array([x4]), hw, 1, 8, 256, hip.ptr(V), None, None, None, hip.ptr(amax), hip.ptr(vinv), hip.stream_ptr()), "lgd_wino_in_h2")
        M = ops._h2_product(lib, "fwd", filt.fwd, 256, 256, V, vinv, False, filt.inv, ops._freq_buf(64, 256, T, dev))
        y = torch.empty_like(x4)
        ops._wino_out(lib, M[:, 0], b4, hw, 1, 8, 256, 6, False, [y], None)
        return filt.fwd, V, M, y
    U, _ = ops._wino_filters(lib, [w4], [None], 256, dev, 6, T, False)
    V = ops._freq_buf(64, 256, T, dev)
    hip.check(lib.lgd_wino_in(hip.ptr_array([x4]), hw, 1, 8, 256, 6, hip.ptr(V), None, None, None, hip.stream_ptr()), "lgd_wino_in")
    M = ops._wino_gemm("wino_gemm_fwd", U, V, out=ops._freq_buf(64, 256, T, dev))
    y = torch.empty_like(x4)
    ops._wino_out(lib, M[:, 0], b4, hw, 1, 8, 256, 6, False, [y], None)
    return (U.t if isinstance(U, ops._FilterImage) else U), V, M, y


# what runs on the side stream: the whole p3 convolution (default) or ONE of its kernels in a loop
AG = sys.argv[sys.argv.index("--aggressor") + 1] if "--aggressor" in sys.argv else "conv"
hw3 = hip.int_array([100, 168])
T3 = lib.lgd_wino_tiles(hw3, 1, 8, 6)
_filt3 = ops._h2_filters(lib, [w3], [None], 256, dev, False)
_amax3 = ops._amax_bits(lib, [x3], hw3)
_V3, _vinv3 = ops._h2_buf(256, T3, dev), torch.empty(1, dtype=torch.float32, device=dev)
hip.check(lib.lgd_wino_in_h2(hip.ptr_array([x3]), hw3, 1, 8, 256, hip.ptr(_V3), None, None, None, hip.ptr(_amax3), hip.ptr(_vinv3), hip.stream_ptr()), "lgd_wino_in_h2")
_M3 = ops._freq_buf(64, 256, T3, dev)
_y3 = torch.empty_like(x3)
_V3f = ops._freq_buf(64, 256, T3, dev).normal_()
_Ug3 = torch.randn((64, 256, 256), device=dev) * 0.05
ops.gemm3_backend(True, force=True)
_W2h = torch.randn((256, 256), device=dev) * 0.05
_x2h = torch.randn((8, 256, 16800), device=dev)
_am2h = _x2h.abs().max().reshape(1).view(torch.int32) + 0
_o2h = torch.empty((8, 256, 16800), device=dev)
_filt3b = ops._h2_filters(lib, [w3[:128].contiguous()], [None], 256, dev, False)
_M3b = ops._freq_buf(64, 128, T3, dev)
_dminv3 = torch.ones(64, device=dev)
torch.cuda.synchronize()
AGGRESSOR = {
    "conv": lambda: ops.conv3x3(x3, w3, None),
    "wino_in_h2": lambda: hip.check(lib.lgd_wino_in_h2(hip.ptr_array([x3]), hw3, 1, 8, 256, hip.ptr(_V3), None, None, None, hip.ptr(_amax3), hip.ptr(_vinv3),
                                                       hip.stream_ptr()), "lgd_wino_in_h2"),
    "wino_in": lambda: hip.check(lib.lgd_wino_in(hip.ptr_array([x3]), hw3, 1, 8, 256, 6, hip.ptr(_V3f), None, None, None, hip.stream_ptr()), "lgd_wino_in"),
    "h2_fwd": lambda: ops._h2_product(lib, "fwd", _filt3.fwd, 256, 256, _V3, _vinv3, False, _filt3.inv, _M3),
    "wino_out": lambda: ops._wino_out(lib, _M3[:, 0], None, hw3, 1, 8, 256, 6, False, [_y3], None),
    "amax_maps": lambda: ops._amax_bits(lib, [x3.clone()], hw3),
    "filters": lambda: ops._h2_filters(lib, [w3], [None], 256, dev, False),
    "gemm3": lambda: ops.gemm3_bmm(_Ug3, _V3f, out=_M3),
    "gemm2h": lambda: ops.gemm2h_bmm(_W2h.view(1, 256, 256).expand(8, 256, 256), _x2h, _am2h, out=_o2h),
    "h2_fwd128": lambda: ops._h2_product(lib, "fwd", _filt3b.fwd, 128, 256, _V3, _vinv3, False, _filt3b.inv, _M3b),
    "h2_dw": lambda: ops._h2_dw(lib, _V3, _dminv3, _V3, _vinv3, 256, 256),
    "copy": lambda: _y3.copy_(x3),
    "none": lambda: None,
}[AG]
if TORCH_ONLY:
    xt = torch.randn((64, 256, 1024), device=dev, generator=g)

    def staged():   # noqa: F811
        y = xt * 1.5 + 2.0                      # producer: 64 MB written by an elementwise kernel
        z = y.sum(dim=2, dtype=torch.float64)   # consumer: a reduction over what it wrote
        w = torch.tanh(y)                       # a second consumer
        return y, z, w, w.sum(dim=0)
    names_override = ("y = x*1.5+2", "sum(y)", "tanh(y)", "sum(tanh(y))")
with torch.no_grad():
    quiet = staged()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    names = names_override if TORCH_ONLY else ("filter image", "V", "M", "y")
    bad = [0, 0, 0, 0]
    detail = None
    for r in range(ROUNDS):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                AGGRESSOR()
        loud = staged()
        torch.cuda.synchronize()
        for i, (q, l) in enumerate(zip(quiet, loud)):
            n = int((q != l).sum())
            bad[i] += n > 0
            if n and detail is None and i >= 1:
                idx = torch.nonzero(q.contiguous() != l.contiguous())
                detail = (names[i], r, n, idx[:6].tolist(), idx[-3:].tolist())
                if i == 3:
                    FIRST_IDX, LOUD_Y = tuple(idx[0].tolist()), l.clone()
    if detail is not None and detail[0] == "y" and not TORCH_ONLY:
        # fingerprint of the first wrong tile: y[i][0] = sum_a AT[i][a] (m[a][0] + ... + m[a][6]) -- which frequency row a, and what did the kernel use for it?
        AT = torch.tensor([[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, .5, -.5, 0], [0, 1, 1, 4, 4, .25, .25, 0], [0, 1, -1, 8, -8, .125, -.125, 0],
                           [0, 1, 1, 16, 16, 1 / 16, 1 / 16, 0], [0, 1, -1, 32, -32, 1 / 32, -1 / 32, 1]], dtype=torch.float64)
        n_, c_, row, col = FIRST_IDX
        ty, tx = row // 6, col // 6
        tile = (n_ * 9 + ty) * 14 + tx
        Mq = quiet[2]                                   # (64, C, T) view
        m = Mq[:, c_, tile].double().cpu().view(8, 8)
        want = (AT @ m @ AT.T)[:, 0] + float(b4[c_])
        got = LOUD_Y[n_, c_, 6 * ty:6 * ty + 6, 6 * tx].double().cpu()
        ref = quiet[3][n_, c_, 6 * ty:6 * ty + 6, 6 * tx].double().cpu()
        print("first wrong tile: image %d channel %d tile (%d, %d) = tile %d of the run; column 0 of its block, rows 0..5:" % (n_, c_, ty, tx, tile))
        print("   quiet kernel   ", ["%.5f" % v for v in ref.tolist()])
        print("   fp64 from M    ", ["%.5f" % v for v in want[: len(ref)].tolist()])
        print("   loud kernel    ", ["%.5f" % v for v in got.tolist()])
        dlt = got - ref
        print("   loud - quiet   ", ["%.5f" % v for v in dlt.tolist()])
        for a_ in range(8):
            col_a = AT[: len(dlt), a_]
            if float(col_a.abs().sum()) == 0:
                continue
            delta = float((dlt * col_a).sum() / (col_a * col_a).sum())
            resid = float((dlt - delta * col_a).abs().max())
            print("   if frequency row a = %d alone were off by d: d = %+.5f, unexplained %.2e   (true m[a][0] = %+.5f, m[a][0] + d = %+.5f)" % (a_, delta, resid, float(m[a_, 0]), float(m[a_, 0]) + delta))
    for n, b in zip(names, bad):
        print("%-14s differs from the quiet run in %2d of %d rounds" % (n, b, ROUNDS))
    print("first difference:", detail)
