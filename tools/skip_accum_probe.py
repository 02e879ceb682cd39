"""Go / no-go probe: the identity-shortcut gradient sum of a bottleneck block (dx_conv1 + g, a separate add pass over the block's
input-size map) folded into conv1's input-gradient GEMM as beta = 1 accumulation -- library conv backward + add vs baddbmm in place.
Shapes: the identity blocks of res3 / res4 / res5 at config 2 (8 x 800 x 1344).  Run with PYTORCH_TUNABLEOP_ENABLED=1
PYTORCH_TUNABLEOP_TUNING=1 to let TunableOp pick the strided-batched solution."""
import torch

dev = "cuda"
shapes = [("res3", 512, 128, 100 * 168), ("res4", 1024, 256, 50 * 84), ("res5", 2048, 512, 25 * 42)]
N = 8


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, Ci, Co, HW in shapes:
    H = int(HW ** 0.5)
    while HW % H:
        H -= 1
    W = HW // H
    x = torch.randn(N, Ci, H, W, device=dev)
    w = torch.randn(Co, Ci, 1, 1, device=dev) * 0.05
    dz = torch.randn(N, Co, H, W, device=dev)
    g = [torch.randn(N, Ci, H, W, device=dev) for _ in range(4)]   # rotate: > 256 MB in flight for res3
    k = [0]

    def conv_add():
        k[0] = (k[0] + 1) % 4
        dx = torch.ops.aten.convolution_backward(dz, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        return dx + g[k[0]]

    def conv_only():
        return torch.ops.aten.convolution_backward(dz, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]

    wT = w.view(Co, Ci).t().contiguous()

    def accum():
        k[0] = (k[0] + 1) % 4
        a = g[k[0]].view(N, Ci, HW)
        return torch.baddbmm(a, wT.unsqueeze(0).expand(N, Ci, Co), dz.view(N, Co, HW), out=a)

    def bmm_only():
        return torch.bmm(wT.unsqueeze(0).expand(N, Ci, Co), dz.view(N, Co, HW))

    t = [timeit(f) for f in (conv_only, conv_add, bmm_only, accum)]
    ref = conv_only() + g[0]
    a0 = g[0].clone()
    got = torch.baddbmm(a0.view(N, Ci, HW), wT.unsqueeze(0).expand(N, Ci, Co), dz.view(N, Co, HW), out=a0.view(N, Ci, HW)).view_as(ref)
    err = float((got - ref).abs().max() / ref.abs().max())
    print("%s Ci=%d Co=%d HW=%d: conv dx %.1f us | conv dx + add %.1f us | bmm %.1f us | baddbmm in place (beta=1) %.1f us | rel err %.1e"
          % (name, Ci, Co, HW, t[0], t[1], t[2], t[3], err))
