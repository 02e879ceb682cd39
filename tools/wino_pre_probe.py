"""lgd_wino_in with and without the folded pre-activation (bias + ReLU on load, 16-bit tile masks written) on the backbone's conv2
input shapes at config 2; buffers rotate so that inputs come from HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lgd_amd import hip

lib = hip.load()
dev = "cuda"
for name, C, H, W in (("res3", 128, 100, 168), ("res4", 256, 50, 84), ("res5", 512, 25, 42), ("res2", 64, 200, 336)):
    N, R = 8, 6
    xs = [torch.randn(N, C, H, W, device=dev) for _ in range(R)]
    hw = hip.int_array([H, W])
    T = lib.lgd_wino_tiles(hw, 1, N, 4)
    Vs = [torch.empty(C, 36, T, device=dev) for _ in range(R)]
    bits = torch.empty(C, T, dtype=torch.int16, device=dev)
    pre = torch.randn(C, device=dev)

    def run(with_pre, with_bits, n=30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for w in range(2):
            if w == 1:
                e0.record()
            for i in range(n):
                hip.check(lib.lgd_wino_in(hip.ptr_array([xs[i % R]]), None, None, hw, 1, N, C, 4, 0, hip.ptr(Vs[i % R]), None,
                                          hip.ptr(pre) if with_pre else None, hip.ptr(bits) if with_bits else None, hip.stream_ptr()), "in")
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    a, b, c = run(False, False), run(True, False), run(True, True)
    gb = (N * C * H * W * 4 + C * 36 * T * 4) / 1e9
    print("%s C=%d %dx%d: plain %.1f us (%.2f TB/s) | pre %.1f us | pre + masks %.1f us" % (name, C, H, W, a, gb / a * 1e6 / 1e3 * 1e-3 * 1e3, b, c))
