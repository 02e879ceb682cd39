"""Round 6 diagnosis: which of this library's product kernels returns different bits when ANOTHER of this library's kernels runs beside it on a
second stream (tools/fpn_race_probe.py: with the FPN fork on, p4's convolution -- Winograd transforms + csrc/gemm3.hip, 1008 tiles -- differed in a
few 128-byte runs of frequency batches f = 0 (mod 8) while the p3 convolution ran on the side stream).  Victims on the main stream: gemm3 (bf16x3),
h2_fwd, the library's bmm; aggressors on the side stream: h2_fwd, a copy, gemm3.  30 rounds each, bit-compared with the quiet result.
    python tools/concurrency_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

import common as cm  # noqa: E402
from lgd_amd import hip, ops  # noqa: E402

dev = "cuda"
lib = hip.load()
g = torch.Generator(device=dev).manual_seed(5)
nb, M, K = 64, 256, 256


def operands(T):
    a = torch.randn((nb, M, K), device=dev, generator=g) * 0.05
    v = torch.randn((K, nb, T), device=dev, generator=g)
    sa, sv = cm.h2_pow2_scale(a.abs().amax((1, 2))), cm.h2_pow2_scale(v.abs().amax((0, 2)))
    return dict(a=a, v=v, img=cm.h2_split_image(a, sa), vs=cm.h2_split_rows(v, sv), ia=(1 / sa).contiguous(), iv=(1 / sv).contiguous(), T=T)


def h2(o, out=None):
    T = o["T"]
    out = torch.empty((M, nb, T), device=dev) if out is None else out
    hip.check(lib.lgd_h2_fwd(hip.ptr(o["img"]), hip.ptr(o["vs"]), 4 * T, 4 * nb * T, 4 * o["vs"].numel(), hip.ptr(out), T, nb * T, hip.ptr(o["ia"]), hip.ptr(o["iv"]), 1,
                             None, nb, M, T, K, hip.stream_ptr()), "lgd_h2_fwd")
    return out


ops.gemm3_backend(True, force=True)
small, big = operands(1024), operands(3808)
_w2 = torch.randn((256, 256), device=dev, generator=g) * 0.05
_x2 = torch.randn((8, 256, 4200), device=dev, generator=g)
_am2 = _x2.abs().max().reshape(1).view(torch.int32) + 0
_dz2 = torch.randn((8, 256, 4200), device=dev, generator=g)
_amd2 = _dz2.abs().max().reshape(1).view(torch.int32) + 0


def pwdw():
    S = lib.lgd_h2_pwdw_splits(8, 256, 256, 4200)
    part = torch.empty((S, 256, 256), device=dev)
    hip.check(lib.lgd_h2_pwdw(hip.ptr(_dz2), hip.ptr(_x2), hip.ptr(_amd2), hip.ptr(_am2), hip.ptr(part), S, 8, 256, 256, 4200, hip.stream_ptr()), "lgd_h2_pwdw")
    return part


victims = {"gemm2h (f16x2, B split in registers: v_fma_mix), 8 x 4200 px": lambda: ops.gemm2h_bmm(_w2.view(1, 256, 256).expand(8, 256, 256), _x2, _am2),
           "h2_pwdw (both operands split in registers)": pwdw,
           "gemm3 (bf16x3), T = 1024": lambda: ops.gemm3_bmm(small["a"], small["v"].permute(1, 0, 2)),
           "h2_fwd, T = 1024": lambda: h2(small),
           "library bmm, T = 1024": lambda: torch.bmm(small["a"], small["v"].permute(1, 0, 2))}
hog_src = torch.empty(1 << 27, dtype=torch.uint8, device=dev)
hog_dst = torch.empty_like(hog_src)
big_out = torch.empty((M, nb, big["T"]), device=dev)
aggressors = {"nothing": lambda: None, "a copy": lambda: hog_dst.copy_(hog_src), "h2_fwd, T = 3808": lambda: h2(big, big_out),
              "gemm3, T = 3808": lambda: ops.gemm3_bmm(big["a"], big["v"].permute(1, 0, 2))}
side = torch.cuda.Stream()
for vn, vf in victims.items():
    quiet = vf()
    torch.cuda.synchronize()
    for an, af in aggressors.items():
        bad = worst = 0
        for _ in range(30):
            with torch.cuda.stream(side):
                for _ in range(3):
                    af()
            outs = [vf() for _ in range(3)]
            torch.cuda.synchronize()
            for o in outs:
                d = (o != quiet)
                n = int(d.sum())
                bad += n > 0
                worst = max(worst, n)
        print("%-66s beside %-18s: %2d of 90 results differ from the quiet one (most elements in one result: %d)" % (vn, an, bad, worst), flush=True)
