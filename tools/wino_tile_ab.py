"""F(4x4,3x3) vs F(6x6,3x3): one 256->256 filter (+ReLU) over the config-2 pyramid (and the 2-image / both-pyramids shapes), forward +
backward, per-kernel HIP-event times and the channel-GEMM times, plus the error of each against the fp64 direct convolution on a
small problem.  usage: python tools/wino_tile_ab.py [B=8]"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgd_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
print("tuned GEMM table:", ops.enable_tuned_gemms())
dev = "cuda"
hws = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
torch.manual_seed(0)

# ---- accuracy on a small problem (fp64 direct convolution on the GPU as the reference)
x = torch.randn(2, 256, 48, 72, device=dev).relu_()
w = torch.randn(256, 256, 3, 3, device=dev) * (2.0 / (9 * 256)) ** 0.5
b = torch.randn(256, device=dev) * 0.1
gy = torch.randn(2, 256, 48, 72, device=dev)
xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
yr = F.conv2d(xr, wr, b.double(), 1, 1)
yr.backward(gy.double())
for tile in (4, 6):
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = ops._Conv3x3.apply(wg, b, False, tile, xg)[0]
    y.backward(gy)
    e = lambda a, r: (float((a.double() - r).abs().max() / r.abs().max()), float((a.double() - r).std() / r.std()))  # noqa: E731
    print("tile %d  fwd max/scale %.2e rms %.2e | dx %.2e %.2e | dw %.2e %.2e" % ((tile,) + e(y.detach(), yr.detach()) + e(xg.grad, xr.grad) + e(wg.grad, wr.grad)))

# ---- timing
res = {}
for name, nb, npyr, Co in (("pyramid 256->256", B, 1, 256), ("both pyramids 256->256", B, 2, 256), ("pyramid 256->720", B, 1, 720)):
    xs = [torch.randn(nb, 256, h, w_, device=dev, requires_grad=True) for _ in range(npyr) for h, w_ in hws]
    w = (torch.randn(Co, 256, 3, 3, device=dev) * 0.02).requires_grad_(True)
    bb = torch.zeros(Co, device=dev, requires_grad=True)
    gys = [torch.randn(nb, Co, h, w_, device=dev) for _ in range(npyr) for h, w_ in hws]
    for tile in [int(t) for t in os.environ.get("LGD_WINO_AB_TILES", "4,6").split(",")]:
        def step():
            ys = ops._Conv3x3.apply(w, bb, True, tile, *xs)
            torch.autograd.backward(ys, gys)
            w.grad = bb.grad = None
            for t in xs:
                t.grad = None
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 10 * 1e3
        ops.kernel_timer_enable(True)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        kt = ops.kernel_timer_collect()
        ops.kernel_timer_enable(False)
        gf = ops.kernel_gemm_flops()
        row = {k: round(1e3 * v[1] / v[0], 1) for k, v in kt.items()}
        tr = sum(v for k, v in row.items() if k.startswith("wino_") and "gemm" not in k and "filter" not in k)
        gm = sum(v for k, v in row.items() if "gemm" in k)
        print("%-24s B=%d tile %d: fwd+bwd %.3f ms wall | transforms %.0f us, GEMMs %.0f us | %s" % (name, nb, tile, wall, tr, gm, json.dumps(row)), flush=True)
        res["%s tile %d" % (name, tile)] = {"wall_ms": wall, "kernels_us": row}
print(json.dumps(res))
