"""The student's 1x1 convolutions in isolation (config 2: B = 8, 800x1344): TFLOP/s of forward / dx / dW as issued by
ops._PointwiseConvBN (F.conv2d, aten.convolution_backward, per-image NT bmm + sum) and as plain batched GEMMs on the NCHW views."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lgd_amd import ops

print("tuned table:", ops.enable_tuned_gemms())
N = 8
SH = [("res3 512->128", 512, 128, 100, 168), ("res3 128->512", 128, 512, 100, 168), ("res4 1024->256", 1024, 256, 50, 84),
      ("res4 256->1024", 256, 1024, 50, 84), ("res5 2048->512", 2048, 512, 25, 42), ("res5 512->2048", 512, 2048, 25, 42),
      ("fpn 512->256", 512, 256, 100, 168), ("fpn 2048->256", 2048, 256, 25, 42), ("res3.0 256->512", 256, 512, 100, 168)]
NSET = 4


def bench(fn, flop, reps=16):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return "%6.0f us %5.0f TF" % (us, flop / us / 1e6)


for name, Ci, Co, H, W in SH:
    xs = [torch.randn(N, Ci, H, W, device="cuda") for _ in range(NSET)]
    dz = [torch.randn(N, Co, H, W, device="cuda") for _ in range(NSET)]
    w = torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05
    w2 = w.view(Co, Ci)
    wt = w2.t().contiguous()
    flop = 2.0 * N * H * W * Ci * Co
    r = {}
    r["conv2d"] = bench(lambda i: F.conv2d(xs[i % NSET], w), flop)
    r["matmul"] = bench(lambda i: torch.matmul(w2, xs[i % NSET].view(N, Ci, -1)), flop)
    r["dx_convbwd"] = bench(lambda i: torch.ops.aten.convolution_backward(dz[i % NSET], xs[i % NSET], w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]), flop)
    r["dx_matmul"] = bench(lambda i: torch.matmul(wt, dz[i % NSET].view(N, Co, -1)), flop)
    r["dw_bmm_sum"] = bench(lambda i: torch.bmm(dz[i % NSET].view(N, Co, -1), xs[i % NSET].view(N, Ci, -1).transpose(1, 2)).sum(0), flop)
    print("%-18s " % name + "  ".join("%s %s" % kv for kv in r.items()), flush=True)
