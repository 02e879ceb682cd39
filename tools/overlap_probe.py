"""Can an HBM-bound Winograd transform run UNDER a library fp32 GEMM (MFMA-bound) on a second stream?
serial = gemm ; transform on one stream, overlap = the same two launches on two streams (no two GEMMs ever concurrent)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lgd_amd import hip, ops  # noqa: E402

lib = hip.load()
ops.enable_tuned_gemms()
dev = torch.device("cuda")
hws = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
s2 = torch.cuda.Stream()


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for maps in (5, 10):
    lv = hws * (maps // 5)
    hw = hip.int_array([d for h in lv for d in h])
    N, C, nf = 8, 256, 36
    T = lib.lgd_wino_tiles(hw, len(lv), N, 4)
    fbuf = lambda: torch.randn(C, nf, T, device=dev).permute(1, 0, 2)  # noqa: E731
    Md, dM, V, Vd = fbuf(), fbuf(), fbuf(), fbuf()
    U = torch.randn(nf, C, C, device=dev)
    dys = [torch.randn(N, C, h, w, device=dev) for h, w in lv]
    dxs = [torch.empty(N, C, h, w, device=dev) for h, w in lv]
    bits = torch.zeros(C, T, dtype=torch.int16, device=dev)
    out_M = torch.empty(C, nf, T, device=dev).permute(1, 0, 2)

    def k_out(stream):
        hip.check(lib.lgd_wino_out(hip.ptr(Md), None, hw, len(lv), N, C, 4, 0, 0, hip.ptr_array(dxs), None, ctypes.c_void_p(stream.cuda_stream)), "out")

    def k_dual(stream):
        hip.check(lib.lgd_wino_in(hip.ptr_array(dys), None, hip.ptr(bits), hw, len(lv), N, C, 4, 0, hip.ptr(Vd), hip.ptr(dM), None, None, ctypes.c_void_p(stream.cuda_stream)), "in")

    def g_dw():
        return torch.bmm(dM, V.transpose(1, 2))

    def g_fwd():
        return torch.bmm(U, V, out=out_M)

    cur = torch.cuda.current_stream()
    for gname, gemm in (("gemm_dw", g_dw), ("gemm_fwd/dx", g_fwd)):
        for kname, kern in (("wino_out", k_out), ("wino_in_dual", k_dual)):
            tg, tk = bench(gemm), bench(lambda: kern(cur))
            ts = bench(lambda: (gemm(), kern(cur)))

            def both():
                s2.wait_stream(cur)
                gemm()
                kern(s2)
                cur.wait_stream(s2)
            to = bench(both)
            print("maps %2d %-12s %.3f ms + %-12s %.3f ms: serial %.3f ms, two streams %.3f ms (hidden %.0f %% of the transform)" % (
                maps, gname, tg, kname, tk, ts, to, 100 * (ts - to) / tk), flush=True)
