"""Winograd conv3x3 vs MIOpen (F.conv2d) fwd / fwd+bwd timings at the LGD path's shapes."""
import time

import torch
import torch.nn.functional as F

from lgd_amd import ops


def t(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for tag, N, Ci, Co, H, W in (("p3", 8, 256, 256, 100, 168), ("p4", 8, 256, 256, 50, 84), ("p5", 8, 256, 256, 25, 42), ("p6", 8, 256, 256, 13, 21), ("p7", 8, 256, 256, 7, 11), ("p5 b2", 2, 256, 256, 25, 42), ("fcos cls", 8, 256, 80, 100, 168),
                             ("p3 cls", 8, 256, 720, 100, 168), ("p3 box", 8, 256, 36, 100, 168), ("res3", 8, 128, 128, 100, 168),
                             ("res4", 8, 256, 256, 50, 84), ("res5", 8, 512, 512, 25, 42)):
    x = torch.randn(N, Ci, H, W, device="cuda", requires_grad=True)
    w = (torch.randn(Co, Ci, 3, 3, device="cuda") * 0.02).requires_grad_(True)
    b = torch.zeros(Co, device="cuda", requires_grad=True)
    gy = torch.randn(N, Co, H, W, device="cuda")
    for name, f in (("miopen", lambda: F.conv2d(x, w, b, 1, 1)), ("wino", lambda: ops._Conv3x3.apply(w, b, False, ops._WINO_TILE, x)[0])):
        with torch.no_grad():
            tf = t(f)

        def fb():
            y = f()
            y.backward(gy)
            x.grad = w.grad = b.grad = None
        tfb = t(fb)
        print("%-7s %-7s fwd %.3f ms  fwd+bwd %.3f ms" % (tag, name, tf, tfb), flush=True)
ops.kernel_timer_enable(True)
x = torch.randn(8, 256, 100, 168, device="cuda", requires_grad=True)
w = (torch.randn(256, 256, 3, 3, device="cuda") * 0.02).requires_grad_(True)
for _ in range(5):
    ops._Conv3x3.apply(w, None, False, ops._WINO_TILE, x)[0].backward(torch.ones(8, 256, 100, 168, device="cuda"))
for k, (n, ms) in sorted(ops.kernel_timer_collect().items()):
    print("%-20s %3d launches  %.1f us avg" % (k, n, ms / n * 1e3))

# the whole pyramid through one filter: per-level library convs vs one concatenated Winograd pass
hws = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
for Co in (256, 720, 36):
    xs = [torch.randn(8, 256, h, w_, device="cuda", requires_grad=True) for h, w_ in hws]
    w = (torch.randn(Co, 256, 3, 3, device="cuda") * 0.02).requires_grad_(True)
    b = torch.zeros(Co, device="cuda", requires_grad=True)
    gys = [torch.randn(8, Co, h, w_, device="cuda") for h, w_ in hws]
    for name, f in (("miopen", lambda: [F.relu(F.conv2d(x, w, b, 1, 1)) for x in xs]), ("wino", lambda: ops._Conv3x3.apply(w, b, True, ops._WINO_TILE, *xs))):
        def fb():
            torch.autograd.backward(f(), gys)
            w.grad = b.grad = None
            for x in xs:
                x.grad = None
        with torch.no_grad():
            tf = t(f)
        print("pyramid 256->%d +relu %-7s fwd %.3f ms  fwd+bwd %.3f ms" % (Co, name, tf, t(fb)), flush=True)
