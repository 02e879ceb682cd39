"""1x1 convolutions of the student: MIOpen (F.conv2d fwd, aten.convolution_backward) vs plain batched GEMMs on NCHW."""
import time

import torch
import torch.nn.functional as F


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for tag, N, Ci, Co, H, W in (("res3 512->128", 8, 512, 128, 100, 168), ("res3 128->512", 8, 128, 512, 100, 168),
                             ("res4 1024->256", 8, 1024, 256, 50, 84), ("res4 256->1024", 8, 256, 1024, 50, 84),
                             ("res5 2048->512", 8, 2048, 512, 25, 42), ("res5 512->2048", 8, 512, 2048, 25, 42),
                             ("lat3 512->256", 8, 512, 256, 100, 168), ("res2 256->64", 8, 256, 64, 200, 336)):
    x = torch.randn(N, Ci, H, W, device="cuda")
    w = torch.randn(Co, Ci, 1, 1, device="cuda") * 0.02
    b = torch.randn(Co, device="cuda")
    gy = torch.randn(N, Co, H, W, device="cuda")
    w2 = w.view(Co, Ci)
    f_mi = t(lambda: F.conv2d(x, w, b))
    f_mm = t(lambda: torch.matmul(w2, x.view(N, Ci, H * W)))
    d_mi = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, [Co], [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
    d_mm = t(lambda: torch.matmul(w2.t(), gy.view(N, Co, H * W)))
    w_mi = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, [Co], [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
    w_mm = t(lambda: torch.bmm(gy.view(N, Co, H * W), x.view(N, Ci, H * W).transpose(1, 2)).sum(0))
    fl = 2.0 * N * Co * Ci * H * W / 1e9
    print("%-15s %5.1f GF | fwd miopen %.3f mm %.3f | dgrad miopen %.3f mm %.3f | wgrad miopen %.3f mm %.3f ms" %
          (tag, fl, f_mi, f_mm, d_mi, d_mm, w_mi, w_mm), flush=True)
